// tile_persist_kernel -- persistent form of the tiled hypercube pass (Full / Parity index spaces), gfx950.
//
// Same tables (plan.h: DevPass + DevQuad records, built by mat.cpp:build_pass) and the same arithmetic per
// record as tile_pass_kernel (matvec_kernels.hip); what differs is the execution shape:
//   * one workgroup per CU stays resident and walks the tiles with a grid stride (tile ids in the same XCD-aware
//     order as the one-tile-per-workgroup launch: the 32 workgroups of an XCD work on one XCD group at a time);
//   * the LDS tile is double-buffered and filled by DMA (global_load_lds, no staging registers): right behind
//     the barrier that publishes tile t, every wave requests its rows of tile t+1 into the other buffer, so that
//     tile's HBM latency runs under the LDS phase and the stores of tile t.  A wave's loads return in order:
//     the next thing the wave waits for -- the gathered records of tile t+1 -- needs that tile's barrier next
//     anyway, so nothing waits for the prefetch that would not have waited for the tile;
//   * one barrier per tile: it publishes tile t and, because a wave reaches it only after its last LDS read of
//     tile t-1, hands the other buffer back for tile t+1;
//   * with the whole register file to itself (8 waves per CU) a wave keeps up to DNM_GF (4) gathered records in
//     flight before it multiplies the first;
//   * addresses are  base (SGPR pair) + 32-bit VGPR byte offset: the element index of row k of a thread is
//     phys(row) = pt ^ pk[k] ^ pb  (thread part, k part, block part; XOR-swizzled layout),
//     the partner of a gathered mask is phys(row) ^ phys(mask): one v_xor per load;
//   * the pass descriptor stays in memory and is read with scalar loads where it is needed.
// Streaming rates on MI355X (tools/stream_probe.hip, 32 B/amp): one tile per workgroup 5.4 TB/s, resident
// workgroups with a prefetch 5.8-5.9.
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

template <int MAXS>
__device__ __forceinline__ uint32_t deposit(uint32_t v, int nseg, const int32_t *off, const int32_t *len,
                                            const int32_t *pos) {
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < MAXS; ++j)
    if (j < nseg) r |= ((v >> off[j]) & ((1u << len[j]) - 1u)) << pos[j];
  return r;
}

// The swizzled layout: element `i` of a vector lives at i ^ sw(i).
struct Swz {
  uint32_t sh, msk;   // sw(v) = ((v >> sh) & msk) << 4
  __device__ __forceinline__ uint32_t phys(uint32_t v) const { return v ^ (((v >> sh) & msk) << 4); }
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

// One loop of the off-diagonal part (traits as apply_records of tile_pass_kernel).
template <int R, int LOGNT, bool KVAR, bool CPLX, bool GATHER, bool K0>
__device__ __forceinline__ void records(const DevQuad *__restrict__ quads, uint32_t b, uint32_t e, double (&ar)[R],
                                        double (&ai)[R], const c128 *tile, const c128 *__restrict__ x,
                                        const c128 *__restrict__ xr, uint32_t xr_xor, const Swz &z, uint32_t pt4,
                                        const uint32_t (&ps)[R], uint32_t tid, uint64_t sbase) {
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
    d2v xv[R];
    if constexpr (GATHER) {
      // lanes whose coefficient vanishes for every owned amplitude fetch nothing; a wavefront with no live
      // lane skips the record
      const bool live = KVAR || (a0 + a1 != 0.0) || (CPLX && (a2 + a3 != 0.0));
      if (!__any(live)) continue;
      const c128 *__restrict__ src = Q.src ? xr : x;
      const uint32_t xm = z.phys(Q.mask_loc) ^ (Q.src ? xr_xor : 0u);   // phys(row ^ mask) = phys(row) ^ phys(mask)
      if (live) {
#pragma unroll
        for (int k = 0; k < R; ++k) xv[k] = *elem(src, pt4, ps[k] ^ xm);
      } else {
#pragma unroll
        for (int k = 0; k < R; ++k) xv[k] = d2v{0.0, 0.0};
      }
    } else if constexpr (K0) {
      const d2v *p = reinterpret_cast<const d2v *>(tile + (tid ^ Q.mask_tile));
#pragma unroll
      for (int k = 0; k < R; ++k) xv[k] = p[k * NT];
    } else {
      const uint32_t mt = Q.mask_tile;
      const uint32_t p_lo = tid ^ (mt & (NT - 1u));
      const uint32_t mk = mt >> LOGNT;
#pragma unroll
      for (int k = 0; k < R; ++k) xv[k] = *reinterpret_cast<const d2v *>(tile + (p_lo + (((uint32_t)k ^ mk) << LOGNT)));
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
}

// ---- gathered real records, several in flight ---------------------------------------------------------------
// A slot holds one record between the issue of its loads and its multiply.  Slots are compile-time objects (the
// pipeline below is unrolled), liveness is decided per tile: a wavefront whose lanes all see a vanishing
// coefficient neither loads nor multiplies.
template <int R>
struct GSlot {
  d2v xv[R];
  double a0, a1;
  uint32_t s0, s1;
  bool live;
};

template <int R, int LOGNT>
__device__ __forceinline__ void slot_issue(GSlot<R> &g, const DevQuad &Q, const c128 *__restrict__ x,
                                           const c128 *__restrict__ xr, uint32_t xr_xor, const Swz &z, uint32_t pt4,
                                           const uint32_t (&ps)[R], uint32_t tid, uint64_t sbase) {
  g.a0 = slot_amp(Q, 0, tid, sbase);
  g.a1 = slot_amp(Q, 1, tid, sbase);
  g.s0 = Q.sign_tile[0] >> LOGNT;
  g.s1 = Q.sign_tile[1] >> LOGNT;
  const bool lane_live = ((g.s0 | g.s1) != 0) || (g.a0 + g.a1 != 0.0);
  g.live = __any(lane_live) != 0;
  if (g.live) {
    const c128 *__restrict__ src = Q.src ? xr : x;
    const uint32_t xm = z.phys(Q.mask_loc) ^ (Q.src ? xr_xor : 0u);   // phys(row ^ mask) = phys(row) ^ phys(mask)
    if (lane_live) {
#pragma unroll
      for (int k = 0; k < R; ++k) g.xv[k] = *elem(src, pt4, ps[k] ^ xm);
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k) g.xv[k] = d2v{0.0, 0.0};
    }
  }
}

template <int R>
__device__ __forceinline__ void slot_apply(const GSlot<R> &g, double (&ar)[R], double (&ai)[R]) {
  if (!g.live) return;
  if ((g.s0 | g.s1) == 0) {
    const double cre = g.a0 + g.a1;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      ar[k] = fma(cre, g.xv[k].x, ar[k]);
      ai[k] = fma(cre, g.xv[k].y, ai[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const double cre = flip_sign(g.a0, (uint32_t)__popc(k & g.s0) & 1u) + flip_sign(g.a1, (uint32_t)__popc(k & g.s1) & 1u);
      ar[k] = fma(cre, g.xv[k].x, ar[k]);
      ai[k] = fma(cre, g.xv[k].y, ai[k]);
    }
  }
}

// records [gb, ge): NS slots (2 or 4), record j+NS is issued when record j has been multiplied
template <int R, int LOGNT, int NS>
__device__ __forceinline__ void gather_pipeline(const DevQuad *__restrict__ quads, uint32_t gb, uint32_t ge,
                                                double (&ar)[R], double (&ai)[R], const c128 *__restrict__ x,
                                                const c128 *__restrict__ xr, uint32_t xr_xor, const Swz &z,
                                                uint32_t pt4, const uint32_t (&ps)[R], uint32_t tid, uint64_t sbase) {
#define DNM_ISSUE(G, Q) if ((Q) < ge) slot_issue<R, LOGNT>(G, quads[(Q)], x, xr, xr_xor, z, pt4, ps, tid, sbase); else G.live = false
  if constexpr (NS == 2) {
    GSlot<R> g0, g1;
    g0.live = g1.live = false;
    for (uint32_t q = gb; q < ge; q += 4u) {
      DNM_ISSUE(g0, q);
      DNM_ISSUE(g1, q + 1u);
      slot_apply<R>(g0, ar, ai);
      DNM_ISSUE(g0, q + 2u);
      slot_apply<R>(g1, ar, ai);
      DNM_ISSUE(g1, q + 3u);
      slot_apply<R>(g0, ar, ai);
      slot_apply<R>(g1, ar, ai);
    }
  } else {
    GSlot<R> g0, g1, g2, g3;
    g0.live = g1.live = g2.live = g3.live = false;
    for (uint32_t q = gb; q < ge; q += 8u) {
      DNM_ISSUE(g0, q);
      DNM_ISSUE(g1, q + 1u);
      DNM_ISSUE(g2, q + 2u);
      DNM_ISSUE(g3, q + 3u);
      slot_apply<R>(g0, ar, ai);
      DNM_ISSUE(g0, q + 4u);
      slot_apply<R>(g1, ar, ai);
      DNM_ISSUE(g1, q + 5u);
      slot_apply<R>(g2, ar, ai);
      DNM_ISSUE(g2, q + 6u);
      slot_apply<R>(g3, ar, ai);
      DNM_ISSUE(g3, q + 7u);
      slot_apply<R>(g0, ar, ai);
      slot_apply<R>(g1, ar, ai);
      slot_apply<R>(g2, ar, ai);
      slot_apply<R>(g3, ar, ai);
    }
  }
#undef DNM_ISSUE
}

}  // namespace

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// REGS: the next tile is prefetched into registers (and written to the other LDS buffer at the top of its
// iteration) instead of by DMA
template <int B, int LOGR, bool REGS>
__global__ void __launch_bounds__(1 << (B - LOGR))
tile_persist_kernel(const DevPass *__restrict__ Pp, const PassCall C, const c128 *__restrict__ x,
                    c128 *__restrict__ y, const c128 *__restrict__ xr, uint32_t nblocks) {
  constexpr int R = 1 << LOGR;
  constexpr int LOGNT = B - LOGR;
  constexpr uint32_t NT = 1u << LOGNT;
  constexpr uint32_t TILE = 1u << B;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c128 *bufs = reinterpret_cast<c128 *>(smem);  // two tiles
  const DevPass &P0 = *Pp;

  const uint32_t tid = threadIdx.x;
  Swz z;
  {
    const int S = P0.swz_shift;
    z.sh = S ? (uint32_t)S : 31u;
    z.msk = S ? ((1u << (S - 4)) - 1u) : 0u;
  }
  const bool need_tile = P0.need_tile != 0;
  uint32_t bid = blockIdx.x;
  if (bid >= nblocks) return;

  // ================= compute waves =================
  // element index of row k of this thread in tile `base`: pt ^ pk[k] ^ phys(base)
  const uint32_t pt4 = z.phys(deposit<MAXSEG>(tid, P0.nseg, P0.seg_off, P0.seg_len, P0.seg_pos)) << 4;
  uint32_t pk[R];
#pragma unroll
  for (int k = 0; k < R; ++k)
    pk[k] = z.phys(deposit<MAXSEG>((uint32_t)k << LOGNT, P0.nseg, P0.seg_off, P0.seg_len, P0.seg_pos));
  double dsum_r = 0.0, dsum_i = 0.0, dsum_n = 0.0;     // fused <x, y>, |y|^2 over this workgroup's tiles
  uint32_t cur = 0;
  const uint32_t wave_base = tid & ~63u;               // LDS-DMA: the wave's 64 lanes land on consecutive slots
  d2v pre[R];
  if (need_tile) {
    const uint32_t pb = z.phys(deposit<MAXBSEG>(bid, P0.nbseg, P0.bseg_off, P0.bseg_len, P0.bseg_pos));
    if constexpr (REGS) {
#pragma unroll
      for (int k = 0; k < R; ++k) pre[k] = *elem(x, pt4, pk[k] ^ pb);
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)elem(x, pt4, pk[k] ^ pb),
                                         (LDS_AS void *)(bufs + (k * NT + wave_base)), 16, 0, 0);
    }
  }

  for (; bid < nblocks; bid += gridDim.x) {
    // the descriptor is re-read through the scalar cache in every iteration: kept live across the loop its fields
    // would not fit in the scalar registers (spills)
    const DevPass *Pl = Pp;
    asm volatile("" : "+s"(Pl));
    const DevPass &P = *Pl;
    const uint32_t base = deposit<MAXBSEG>(bid, P.nbseg, P.bseg_off, P.bseg_len, P.bseg_pos);
    const uint64_t sbase = P.sign_base | (uint64_t)base;
    const uint32_t pb = z.phys(base);
    uint32_t ps[R];
#pragma unroll
    for (int k = 0; k < R; ++k) ps[k] = pk[k] ^ pb;
    const uint32_t yx = P.swz_xor_y, xrx = P.swz_xor_src;
    const bool has_diag = P.has_diag != 0, accumulate = P.accumulate != 0;
    const DevQuad *__restrict__ quads = P.quads;
    const c128 *tile = bufs + cur * TILE;
    if constexpr (REGS) {
      if (need_tile) {      // (the waves still reading are on the OTHER buffer: no barrier needed before these writes)
#pragma unroll
        for (int k = 0; k < R; ++k) *reinterpret_cast<d2v *>(bufs + cur * TILE + (tid + k * NT)) = pre[k];
      }
    }

    // ---- accumulator start values (streamed: read once)
    double ar[R], ai[R];
    if (accumulate) {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const d2v v = __builtin_nontemporal_load(elem((const c128 *)y, pt4, ps[k] ^ yx));
        ar[k] = v.x;
        ai[k] = v.y;
      }
    } else if (C.zinit) {
      const c128 *__restrict__ zv = (const c128 *)C.zinit;
      const double zs = -C.zscale;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const d2v v = __builtin_nontemporal_load(elem(zv, pt4, ps[k] ^ yx));
        ar[k] = zs * v.x;
        ai[k] = zs * v.y;
      }
      if (C.zinit2) {
        const c128 *__restrict__ z2 = (const c128 *)C.zinit2;
        const double cr = C.z2re, ci = C.z2im;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const d2v v = __builtin_nontemporal_load(elem(z2, pt4, ps[k] ^ yx));
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

    // ---- gathered records: the workgroups of an XCD are in the same iteration and ask for the same lines
    // within the same microseconds, so the L2 merges the requests
#define DNM_GLOOP(LP, KV, CX) \
  records<R, LOGNT, KV, CX, true, false>(quads, P.loop[LP], P.loop[LP + 1], ar, ai, tile, x, xr, xrx, z, pt4, ps, tid, sbase)
    gather_pipeline<R, LOGNT, (NT >= 1024 || REGS) ? 2 : 4>(quads, P.loop[LP_GATHER_REAL], P.loop[LP_GATHER_KVAR_REAL + 1], ar, ai, x, xr, xrx, z, pt4,
                              ps, tid, sbase);
    DNM_GLOOP(LP_GATHER_CPLX, false, true);
    DNM_GLOOP(LP_GATHER_KVAR_CPLX, true, true);
#undef DNM_GLOOP

    // ---- diagonal, part 1: terms whose sign mask lies outside the tile (one per lane, butterfly sum)
    double dext = 0.0;
    if (has_diag) {
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
    if (need_tile) {
      if constexpr (!REGS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows of the tile have landed
      __syncthreads();                                   // tile `bid` is complete; the other buffer is free
      const uint32_t nb = bid + gridDim.x;
      if (nb < nblocks) {                                // tile t+1 is requested under the LDS phase of tile t
        const uint32_t pbn = z.phys(deposit<MAXBSEG>(nb, P.nbseg, P.bseg_off, P.bseg_len, P.bseg_pos));
        if constexpr (REGS) {
#pragma unroll
          for (int k = 0; k < R; ++k) pre[k] = *elem(x, pt4, pk[k] ^ pbn);
        } else {
          c128 *dst = bufs + (cur ^ 1u) * TILE;
#pragma unroll
          for (int k = 0; k < R; ++k)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)elem(x, pt4, pk[k] ^ pbn),
                                             (LDS_AS void *)(dst + (k * NT + wave_base)), 16, 0, 0);
        }
      }
    }

    // ---- diagonal, part 2: terms that see the tile, bucketed by their k-bit pattern, Walsh-Hadamard over k
    if (has_diag) {
      double D[R];
#pragma unroll
      for (int j = 0; j < R; ++j) D[j] = 0.0;
      D[0] = dext;
      const double *__restrict__ dt = P.dtile;
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
      if (dt) {
#pragma unroll
        for (int k = 0; k < R; ++k) D[k] += dt[tid + k * NT];
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const c128 xs = tile[tid + k * NT];
        ar[k] = fma(D[k], xs.x, ar[k]);
        ai[k] = fma(D[k], xs.y, ai[k]);
      }
    }

    // ---- masks inside the tile
#define DNM_TLOOP(LP, KV, CX, KZ) \
  records<R, LOGNT, KV, CX, false, KZ>(quads, P.loop[LP], P.loop[LP + 1], ar, ai, tile, x, xr, xrx, z, pt4, ps, tid, sbase)
    DNM_TLOOP(LP_TILE_REAL_K0, false, false, true);
    DNM_TLOOP(LP_TILE_REAL, false, false, false);
    DNM_TLOOP(LP_TILE_CPLX, false, true, false);
    DNM_TLOOP(LP_TILE_KVAR_REAL, true, false, false);
    DNM_TLOOP(LP_TILE_KVAR_CPLX, true, true, false);
#undef DNM_TLOOP

#pragma unroll
    for (int k = 0; k < R; ++k) __builtin_nontemporal_store(d2v{ar[k], ai[k]}, elem(y, pt4, ps[k] ^ yx));

    if (C.dot_out) {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const c128 xs = tile[tid + k * NT];
        dsum_r = fma(xs.x, ar[k], dsum_r);
        dsum_r = fma(xs.y, ai[k], dsum_r);
        dsum_i = fma(xs.x, ai[k], dsum_i);
        dsum_i = fma(-xs.y, ar[k], dsum_i);
        dsum_n = fma(ar[k], ar[k], dsum_n);
        dsum_n = fma(ai[k], ai[k], dsum_n);
      }
    }
    cur ^= 1u;
  }

  // ---- fused <x, y> (Lanczos alpha) and |y|^2: one partial per wavefront (summed by the reduction kernel)
  if (C.dot_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      dsum_r += __shfl_xor(dsum_r, off, 64);
      dsum_i += __shfl_xor(dsum_i, off, 64);
      dsum_n += __shfl_xor(dsum_n, off, 64);
    }
    if ((tid & 63u) == 0) {
      const size_t slot = (size_t)blockIdx.x * (NT / 64u) + (tid >> 6);
      C.dot_out[3 * slot] = dsum_r;
      C.dot_out[3 * slot + 1] = dsum_i;
      C.dot_out[3 * slot + 2] = dsum_n;
    }
  }
}

// ---------------------------------------------------------------------------
bool tile_persist_supported(int B, int logR) {
  switch (B * 16 + logR) {
    case 8 * 16 + 2: case 10 * 16 + 2: case 10 * 16 + 3: case 11 * 16 + 3: case 12 * 16 + 3: case 12 * 16 + 4:
    case 12 * 16 + 2: case 11 * 16 + 2:
      return true;
  }
  return false;
}

// workgroups of the persistent launch: one per CU (two LDS tiles of B = 12 each), fewer when the pass has fewer
// tiles; DNM_PERSIST_WGS_PER_CU overrides (experiments)
unsigned tile_persist_grid(int n_loc, int B) {
  static const int per_cu = []() {
    const char *e = getenv("DNM_PERSIST_WGS_PER_CU");
    return e ? atoi(e) : 1;
  }();
  static const int cus = []() {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount;
    }
    return n;
  }();
  const uint64_t tiles = (uint64_t)1 << (n_loc - B);
  return (unsigned)std::min<uint64_t>(tiles, (uint64_t)per_cu * (uint64_t)cus);
}

size_t tile_persist_dot_partials(int n_loc, int B, int logR) {
  return (size_t)tile_persist_grid(n_loc, B) * (((size_t)1 << (B - logR)) / 64);
}

template <int B, int LOGR>
static int launch_persist_cfg(const DevPass *P_dev, const PassCall &call, int n_loc, const void *x, void *y,
                              const void *xr, hipStream_t st) {
  constexpr int NT = 1 << (B - LOGR);
  const size_t lds = (size_t)32 << B;              // double-buffered tile
  const unsigned nblocks = 1u << (n_loc - B);
  const unsigned grid = tile_persist_grid(n_loc, B);
  static const bool regs = []() { const char *e = getenv("DNM_PERSIST_STAGE"); return !(e && e[0] == 'd'); }();
  auto k = regs ? tile_persist_kernel<B, LOGR, true> : tile_persist_kernel<B, LOGR, false>;
  static bool attr_done = false;
  if (!attr_done) {
    DNM_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, P_dev, call, (const c128 *)x, (c128 *)y, (const c128 *)xr,
                     nblocks);
  DNM_HIP(hipGetLastError());
  return 0;
}

// P_dev: the pass descriptor in device memory (its per-call fields are ignored: they travel in `call`)
int launch_tile_persist(const DevPass *P_dev, const PassCall &call, int B, int logR, int n_loc, const void *x, void *y,
                        const void *xr, hipStream_t st) {
  DNM_CHECK(n_loc >= B, "tile larger than the local vector");
  switch (B * 16 + logR) {
    case 8 * 16 + 2: return launch_persist_cfg<8, 2>(P_dev, call, n_loc, x, y, xr, st);
    case 10 * 16 + 2: return launch_persist_cfg<10, 2>(P_dev, call, n_loc, x, y, xr, st);
    case 10 * 16 + 3: return launch_persist_cfg<10, 3>(P_dev, call, n_loc, x, y, xr, st);
    case 11 * 16 + 3: return launch_persist_cfg<11, 3>(P_dev, call, n_loc, x, y, xr, st);
    case 12 * 16 + 3: return launch_persist_cfg<12, 3>(P_dev, call, n_loc, x, y, xr, st);
    case 12 * 16 + 4: return launch_persist_cfg<12, 4>(P_dev, call, n_loc, x, y, xr, st);
    case 12 * 16 + 2: return launch_persist_cfg<12, 2>(P_dev, call, n_loc, x, y, xr, st);
    case 11 * 16 + 2: return launch_persist_cfg<11, 2>(P_dev, call, n_loc, x, y, xr, st);
  }
  set_error("unsupported tile configuration B=%d logR=%d (persistent kernel)", B, logR);
  return 1;
}

}  // namespace dnm
