// SpinConserve(L,k) x SpinConserve(L,k) multiply in the three-field internal layout (sc3.h) for operators whose
// off-diagonal terms are pair hops on ANY bond graph (the kagome Heisenberg model of the reference's flagship example,
// examples/scripts/kagome/run_kagome.py:20-77; long-range exchange; ladders ...): the two tiled passes of
// sc3_kernels.hip with the chain's "bond b couples spins b, b + 1" replaced by a hop table (Sc3Hop).
// Semantics: MatMult_CPU_General (src/dynamite/_backend/bpetsc_template_2.c:371-412) with the index maps of
// bsubspace_impl.h:187-245.
//
// With state = [T | W | Lo] a hop between spins i < j is applied
//   both in Lo      lo pass, from its LDS tile: the partner column is the rank of (pattern ^ mask), looked up in the two
//                   halves of lo_rank that the workgroup keeps in LDS (Sc3Tab::lo_rlo / lo_rhi);
//   both in W       window pass, from its LDS tile: the partner row comes from the operator's table (Sc3Op::wnb);
//   i in Lo, j above   lo pass, gathered: ONE other row (T', W'), the column again by rank -- ascending with the
//                   entry, so a wavefront's loads stay inside a few lines of that row;
//   both above Lo   window pass, gathered: one other block / class at a uniform offset (both in T) or with the rows
//                   remapped (i in W, j in T), the columns as they are.
// Which ket a hop acts on is one popcount: popcount(ket & mask) == |mask| / 2 keeps the number of down spins.
#include "sc3.h"

#include <algorithm>
#include <map>

#include "dnm_common.h"
#include "kernels.h"
#include "sc3_dev.h"

#ifndef DNM_SC3G_PROBE_STAGE
#define DNM_SC3G_PROBE_STAGE 0
#endif

namespace dnm {

namespace {

// what a lane holds of the gathered hop it evaluated for its workgroup's row
struct HopEval {
  int act;
  int64_t delta;
  double c0, c1;
  int need;
  uint32_t xm;
  int rev;
};

// ---------------------------------------------------------------------------------------------------------
// lo pass: y (+)= (hops inside Lo, hops between Lo and the fields above, the diagonal) x.  Workgroup shape, sub-groups
// of 2^m rows, diagonal modes, ACC and the fused sums exactly as sc3_lo_pass (sc3_kernels.hip).
template <int A, int NT, int DIAGM, bool SYM, bool ACC>
__global__ void __launch_bounds__(NT, sc3_win_waves(NT, (sc3_lo_cap(A, NT) * 16 + 1023) / 1024 + 4))
sc3g_lo_pass(const Sc3Tab S, const Sc3Op O, const uint32_t *__restrict__ perm, const Sc3Call C,
             const c128 *__restrict__ xw, c128 *__restrict__ y) {
  constexpr int MAXROWS = cbinom(A, A / 2);
  constexpr int RPT = (MAXROWS + NT - 1) / NT;
  constexpr int H = A / 2, HB = A - H, HS = ilog2c(H + 1);
  constexpr uint32_t HM = (1u << H) - 1u;
  static_assert((1 << HS) == H + 1, "the rows of lo_rhi must be a power of two long (a = 6, 14)");
  // The rank of a flipped pattern costs: byte offsets of its two halves into the LDS tables, two 16-bit reads, one add.
  // The tables hold the ranks times 16 (the byte offset of a tile entry); where registers allow (no on-the-fly diagonal)
  // an entry keeps the two byte offsets of its own pattern and a hop flips them with its own (XOR commutes with the
  // shifts): 10 vector instructions per (entry, hop) where the first form of this loop had 16 (round 5, lab notes).
  constexpr bool HOIST = DIAGM != 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint16_t rka[1 << H];
  __shared__ uint16_t rkb[(1 << HB) * (H + 1)];
  __shared__ double red[3 * (NT / 64)];
  __shared__ double dsh[5 * 8];
  const uint32_t e0 = SC3_CP(uint32_t, perm)[8 * (size_t)blockIdx.x];
  if (e0 == 0xffffffffu) return;
  // the two halves of lo_rank: asked for first, in LDS before the first gathered hop needs them
  uint16_t tka[((1 << H) + NT - 1) / NT], tkb[((1 << HB) * (H + 1) + NT - 1) / NT];
#pragma unroll
  for (int i = 0; i < ((1 << H) + NT - 1) / NT; ++i) {
    const int tt = threadIdx.x + i * NT;
    tka[i] = tt < (1 << H) ? S.lo_rlo[tt] : (uint16_t)0;
  }
#pragma unroll
  for (int i = 0; i < ((1 << HB) * (H + 1) + NT - 1) / NT; ++i) {
    const int tt = threadIdx.x + i * NT;
    tkb[i] = tt < (1 << HB) * (H + 1) ? S.lo_rhi[tt] : (uint16_t)0;
  }
  const int lane = threadIdx.x & 63;
  const int w = S.w;
  const int logm = (int)(e0 >> 30);
  const int NTS = NT >> logm;
  const int sub = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >> (ilog2c(NT) - 6 - logm);
  const int tsub = (int)threadIdx.x & (NTS - 1);
  const uint32_t e = SC3_CP(uint32_t, perm)[8 * (size_t)blockIdx.x + sub];
  const bool has_row = !(e & SC3_NOROW);
  c128 *xs = reinterpret_cast<c128 *>(smem) + (size_t)sub * ((NT * RPT) >> logm);
  const RowId R = decode_row(has_row ? (e & (SC3_NOROW - 1u)) : (e0 & (SC3_NOROW - 1u)), S);
  const uint32_t T = R.T, W = R.W;
  const int kl = R.kl, nrows = has_row ? R.nrows : 0, p = has_row ? R.pitch : 0;
  const int64_t base = R.base;
  const c128 *__restrict__ x = xw - C.win_start;
  const int64_t lbase = base - C.row0;

  SC3_PRIO_MEM();
  uint32_t lowb[RPT];
  c128 xv[RPT];
  const auto pat = SC3_CP(uint16_t, S.lo_pat) + S.lo_off[kl];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = tsub + i * NTS;
    lowb[i] = 0;
    xv[i] = make_double2(0.0, 0.0);
    if (r < nrows) {
      lowb[i] = pat[r];
      xv[i] = x[base + r];
    }
  }
  // gathered hops, one per lane: spin i in Lo, spin j in W or T.  The hop couples this row to ONE other row (T', W');
  // it acts on the entries whose Lo bit has the value the row's bit j leaves open.
  HopEval g{0, 0, 0.0, 0.0, 0, 0u, 0};
  if (lane < O.ngatA) {
    const Sc3Hop h = O.gatA[lane];
    const uint32_t T2 = T ^ h.mT, W2 = W ^ h.mW;
    const int nd = h.half - __popc(T & h.mT) - __popc(W & h.mW);
    const int nlo = __popc(h.mLo);
    const int kr2 = S.k - __popc(T2), cw2 = __popc(W2), kl2 = kr2 - cw2;
    if (nd >= 0 && nd <= nlo && kl2 >= 0 && kl2 <= A && cw2 <= w) {
      const int64_t tb2 = S.ibase[T2];
      if (tb2 >= 0) {
        g.act = 1;
        g.delta = tb2 + S.icoff[kr2 * (w + 1) + cw2] + (int64_t)S.w_rank[W2] * S.pitch[kl2] - base;
        g.need = nd;
        g.xm = h.mLo;
        const bool up = h.dfield == 3 || nd == 1;       // the direction bit is the Lo spin of the pair
        g.c0 = up ? h.up_re : h.dn_re;
        g.c1 = up ? h.up_im : h.dn_im;
      }
    }
  }
  uint64_t hb = has_row ? __ballot(g.act) : 0ull;

#pragma unroll
  for (int i = 0; i < ((1 << H) + NT - 1) / NT; ++i) {
    const int tt = threadIdx.x + i * NT;
    if (tt < (1 << H)) rka[tt] = (uint16_t)(tka[i] << 4);
  }
#pragma unroll
  for (int i = 0; i < ((1 << HB) * (H + 1) + NT - 1) / NT; ++i) {
    const int tt = threadIdx.x + i * NT;
    if (tt < (1 << HB) * (H + 1)) rkb[tt] = (uint16_t)(tkb[i] << 4);
  }
  const unsigned char *rkab = reinterpret_cast<const unsigned char *>(rka), *rkbb = reinterpret_cast<const unsigned char *>(rkb);
  uint32_t elo[HOIST ? RPT : 1], ehi[HOIST ? RPT : 1];
  if constexpr (HOIST) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      elo[i] = (lowb[i] & HM) << 1;
      ehi[i] = (lowb[i] >> H) << (HS + 1);
    }
  }
  // byte offset of the tile entry of (pattern of entry i) ^ (mask m; its halves' offsets mlo, mhi)
  auto flipped = [&](int i, uint32_t m, uint32_t mlo, uint32_t mhi) -> uint32_t {
    uint32_t t, u;
    if constexpr (HOIST) {
      t = elo[i] ^ mlo;
      u = ehi[i] ^ mhi;
    } else {
      const uint32_t p2 = lowb[i] ^ m;
      t = (p2 & HM) << 1;
      u = (p2 >> H) << (HS + 1);
    }
    return (uint32_t)*reinterpret_cast<const uint16_t *>(rkab + t) +
           (uint32_t)*reinterpret_cast<const uint16_t *>(rkbb + (u | ((uint32_t)__popc(t) << 1)));
  };
  // on-the-fly diagonal: per-row sums by the first wavefront of each sub-group (as sc3_lo_pass)
  if (DIAGM == 2 && tsub < 64) {
    const uint64_t hi = ((uint64_t)T << w) | W;
    double v0 = 0.0, vm[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t0 = 0; t0 < O.ndt; t0 += 64) {
      const int t = t0 + lane;
      if (t < O.ndt) {
        const uint64_t sg = SC3_CP(uint64_t, O.dt_sign)[t];
        const double c = flip(SC3_CP(double, O.dt_coef)[t], (uint32_t)__popcll(hi & sg & 0x1fffffffffffffffull) & 1u);
        const int gq = (int)(sg >> 61);
        if (gq == 0) v0 += c;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (gq == j + 1) vm[j] += c;
      }
    }
    v0 = wave_sum(v0);
    if (lane == 0) dsh[5 * sub] = v0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < O.ngroups) {
        const double s = wave_sum(vm[j]);
        if (lane == 0) dsh[5 * sub + j + 1] = s;
      }
  }
  double accr[RPT], acci[RPT];
  // the zero entry behind the row's entries, where the partner table points for hops that do not act
  if (O.ptab && has_row && tsub == 0) xs[nrows] = make_double2(0.0, 0.0);
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = tsub + i * NTS;
    accr[i] = 0.0;
    acci[i] = 0.0;
    if (r < nrows) {
      xs[r] = xv[i];
      if (ACC) {                   // the accumulators start from what the window pass wrote (no registers for y later)
        const c128 yo = load_nt(y + lbase + r);
        accr[i] = yo.x;
        acci[i] = yo.y;
      }
      if (DIAGM == 1) {
        const double dg = __builtin_nontemporal_load(O.diag + lbase + r);
        accr[i] = fma(dg, xv[i].x, accr[i]);
        acci[i] = fma(dg, xv[i].y, acci[i]);
      }
    }
  }
  __syncthreads();                 // the rank tables (and the tile) are in LDS; loads in flight are not waited for
  while (hb) {
    const int m = __ffsll((long long)hb) - 1;
    hb &= hb - 1;
    const c128 *__restrict__ pp = x + (base + rl_i64(g.delta, m));
    const double cr = rl_f64(g.c0, m), ci = rl_f64(g.c1, m);
    const int nd = rl_i32(g.need, m);
    const uint32_t xm = (uint32_t)rl_i32((int)g.xm, m);
    const uint32_t xlo = (xm & HM) << 1, xhi = (xm >> H) << (HS + 1);
    c128 v[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = tsub + i * NTS;
      v[i] = make_double2(0.0, 0.0);
      if (r < nrows && __popc(lowb[i] & xm) == nd)
        v[i] = *reinterpret_cast<const c128 *>(reinterpret_cast<const unsigned char *>(pp) + flipped(i, xm, xlo, xhi));
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      accr[i] = fma(cr, v[i].x, accr[i]);
      acci[i] = fma(cr, v[i].y, acci[i]);
      if (!SYM) {
        accr[i] = fma(-ci, v[i].y, accr[i]);
        acci[i] = fma(ci, v[i].x, acci[i]);
      }
    }
  }
  double dlv[RPT];
  if (DIAGM == 2) {
    const auto dl = SC3_CP(double, O.dlo) + S.lo_off[kl];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = tsub + i * NTS;
      dlv[i] = r < nrows ? dl[r] : 0.0;
    }
  }
  SC3_PRIO_LDS();
  if (DIAGM == 2) {
    const double dg0 = dsh[5 * sub];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = tsub + i * NTS;
      if (r < nrows) {
        double dg = dlv[i] + dg0;
        for (int j = 0; j < O.ngroups; ++j) dg += flip(dsh[5 * sub + j + 1], (uint32_t)__popc(lowb[i] & O.glo[j]) & 1u);
        const c128 xo = xs[r];
        accr[i] = fma(dg, xo.x, accr[i]);
        acci[i] = fma(dg, xo.y, acci[i]);
      }
    }
  }
  // hops inside Lo: the partner column is the rank of the flipped pattern -- from the partner table (Sc3Op::ptab: the
  // table row of an entry in 16-byte words straight from the L2, then one extract, one LDS read and the FMAs per hop)
  if (O.ptab) {
    constexpr int NQ = SC3G_MAX_PTAB / 8;
    const int nq = O.nhp >> 3;
    const uint4 *__restrict__ trow = reinterpret_cast<const uint4 *>(O.ptab + (size_t)O.ptab_row[kl] * (size_t)O.nhp);
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(xs);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = tsub + i * NTS;
      if (r < nrows) {
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
          if (c < nq) {
            const uint4 q0 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(trow) +
                                                              ((uint32_t)r * ((uint32_t)nq * 16u) + 16u * c));
            const uint32_t d0[4] = {q0.x, q0.y, q0.z, q0.w};
            if (SYM) {
              // two hops at a time, no branch among them (the table's last word is padded with hops of coefficient zero
              // that point at the zero entry): their LDS reads go out together
#pragma unroll
              for (int g2 = 0; g2 < 4; ++g2) {
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t ta = d0[g2] & 0xffffu, tb = d0[g2] >> 16;
                const c128 va = *reinterpret_cast<const c128 *>(xb + ta), vb = *reinterpret_cast<const c128 *>(xb + tb);
                const double ua = SC3_CP(double, O.pcoef)[8 * c + 2 * g2], ub = SC3_CP(double, O.pcoef)[8 * c + 2 * g2 + 1];
                accr[i] = fma(ua, va.x, accr[i]);
                acci[i] = fma(ua, va.y, acci[i]);
                accr[i] = fma(ub, vb.x, accr[i]);
                acci[i] = fma(ub, vb.y, acci[i]);
              }
            } else {
#pragma unroll
              for (int hh = 0; hh < 8; ++hh) {
                const int h = 8 * c + hh;
                if (h < O.nldsA) {
                  const auto hp = SC3_CP(Sc3Hop, O.ldsA) + h;
                  const uint32_t t = (hh & 1) ? d0[hh >> 1] >> 16 : d0[hh >> 1] & 0xffffu;
                  const c128 xp = *reinterpret_cast<const c128 *>(xb + t);
                  const bool up = (lowb[i] >> hp->dbit) & 1u;
                  const double cre = up ? hp->up_re : hp->dn_re, cim = up ? hp->up_im : hp->dn_im;
                  accr[i] = fma(cre, xp.x, accr[i]);
                  acci[i] = fma(cre, xp.y, acci[i]);
                  accr[i] = fma(-cim, xp.y, accr[i]);
                  acci[i] = fma(cim, xp.x, acci[i]);
                }
              }
            }
          }
        }
      }
    }
  } else
  for (int hq = 0; hq < O.nldsA; ++hq) {
    const auto hp = SC3_CP(Sc3Hop, O.ldsA) + hq;
    const uint32_t m = hp->mLo;
    const uint32_t mlo = (m & HM) << 1, mhi = (m >> H) << (HS + 1);
    const int half = hp->half, dbit = hp->dbit;
    const double ure = hp->up_re, uim = hp->up_im, dre = hp->dn_re, dim_ = hp->dn_im;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = tsub + i * NTS;
      if (r < nrows && __popc(lowb[i] & m) == half) {
        const c128 xp = *reinterpret_cast<const c128 *>(reinterpret_cast<const unsigned char *>(xs) + flipped(i, m, mlo, mhi));
        if (SYM) {
          accr[i] = fma(ure, xp.x, accr[i]);
          acci[i] = fma(ure, xp.y, acci[i]);
        } else {
          const bool up = (lowb[i] >> dbit) & 1u;
          const double cre = up ? ure : dre, cim = up ? uim : dim_;
          accr[i] = fma(cre, xp.x, accr[i]);
          acci[i] = fma(cre, xp.y, acci[i]);
          accr[i] = fma(-cim, xp.y, accr[i]);
          acci[i] = fma(cim, xp.x, acci[i]);
        }
      }
    }
  }
  double dr = 0.0, di = 0.0, dn = 0.0;
  SC3_PRIO_MEM();
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = tsub + i * NTS;
    if (r < p) {                                  // the padding of a row is written too (zeros)
      double ar = accr[i], ai = acci[i];
      if (r < nrows) {
        if (!ACC && C.zinit) {
          const c128 zv = C.zinit[lbase + r];
          ar = fma(-C.zscale, zv.x, ar);
          ai = fma(-C.zscale, zv.y, ai);
          if (C.zinit2) {
            const c128 z2 = C.zinit2[lbase + r];
            ar = fma(C.z2re, z2.x, ar);
            ar = fma(-C.z2im, z2.y, ar);
            ai = fma(C.z2re, z2.y, ai);
            ai = fma(C.z2im, z2.x, ai);
          }
        }
        if (ACC && C.dot_out) {
          const c128 xo = xs[r];
          dr = fma(xo.x, ar, dr);
          dr = fma(xo.y, ai, dr);
          di = fma(xo.x, ai, di);
          di = fma(-xo.y, ar, di);
          dn = fma(ar, ar, dn);
          dn = fma(ai, ai, dn);
        }
      }
      store_nt(y + lbase + r, ar, ai);
    }
  }
  if (ACC && C.dot_out) {
    dr = wave_sum(dr); di = wave_sum(di); dn = wave_sum(dn);
    if (lane == 0) {
      red[3 * (threadIdx.x >> 6)] = dr;
      red[3 * (threadIdx.x >> 6) + 1] = di;
      red[3 * (threadIdx.x >> 6) + 2] = dn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double sr = 0.0, si = 0.0, sn = 0.0;
      for (int wv = 0; wv < NT / 64; ++wv) { sr += red[3 * wv]; si += red[3 * wv + 1]; sn += red[3 * wv + 2]; }
      C.dot_out[3 * (size_t)blockIdx.x] = sr;
      C.dot_out[3 * (size_t)blockIdx.x + 1] = si;
      C.dot_out[3 * (size_t)blockIdx.x + 2] = sn;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// lo pass on REAL vectors (DNM_MAT_REAL_PACKED): x and y are arrays of doubles in the same positions of the layout; a
// thread owns PAIRS of adjacent row entries, the tile in LDS is an array of doubles -- the shape of sc3_lo_pass_r
// (sc3_kernels.hip) with the hop tables of this file.  Real operators have equal `up` and `dn` elements.
template <int A, int NT, int PPT, int DIAGM, bool ACC>
__global__ void __launch_bounds__(NT, sc3_win_waves(NT, (NT * 2 * PPT * 8 + 1023) / 1024 + 4))
sc3g_lo_pass_r(const Sc3Tab S, const Sc3Op O, const uint32_t *__restrict__ perm, const Sc3Call C,
               const double *__restrict__ xw, double *__restrict__ y) {
  constexpr int MAXROWS = cbinom(A, A / 2);
  constexpr int EPT = 2 * PPT;
  constexpr int H = A / 2, HB = A - H, HS = ilog2c(H + 1);
  constexpr uint32_t HM = (1u << H) - 1u;
  static_assert((1 << HS) == H + 1, "the rows of lo_rhi must be a power of two long (a = 6, 14)");
  static_assert(NT * EPT >= MAXROWS, "the longest row does not fit the workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint16_t rka[1 << H];
  __shared__ uint16_t rkb[(1 << HB) * (H + 1)];
  __shared__ double red[3 * (NT / 64)];
  __shared__ double dsh[5 * 8];
  const uint32_t e0 = SC3_CP(uint32_t, perm)[8 * (size_t)blockIdx.x];
  if (e0 == 0xffffffffu) return;
  uint16_t tka[((1 << H) + NT - 1) / NT], tkb[((1 << HB) * (H + 1) + NT - 1) / NT];
#pragma unroll
  for (int i = 0; i < ((1 << H) + NT - 1) / NT; ++i) {
    const int tt = threadIdx.x + i * NT;
    tka[i] = tt < (1 << H) ? S.lo_rlo[tt] : (uint16_t)0;
  }
#pragma unroll
  for (int i = 0; i < ((1 << HB) * (H + 1) + NT - 1) / NT; ++i) {
    const int tt = threadIdx.x + i * NT;
    tkb[i] = tt < (1 << HB) * (H + 1) ? S.lo_rhi[tt] : (uint16_t)0;
  }
  const int lane = threadIdx.x & 63;
  const int w = S.w;
  const int logm = (int)(e0 >> 30);
  const int NTS = NT >> logm;
  const int sub = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >> (ilog2c(NT) - 6 - logm);
  const int tsub = (int)threadIdx.x & (NTS - 1);
  const uint32_t e = SC3_CP(uint32_t, perm)[8 * (size_t)blockIdx.x + sub];
  const bool has_row = !(e & SC3_NOROW);
  double *xs = reinterpret_cast<double *>(smem) + (size_t)sub * ((NT * EPT) >> logm);
  const RowId R = decode_row(has_row ? (e & (SC3_NOROW - 1u)) : (e0 & (SC3_NOROW - 1u)), S);
  const uint32_t T = R.T, W = R.W;
  const int kl = R.kl, nrows = has_row ? R.nrows : 0, p = has_row ? R.pitch : 0;
  const int64_t base = R.base;
  const double *__restrict__ x = xw - C.win_start;
  const int64_t lbase = base - C.row0;
  // entry i of a thread is column tsub + i * NTS: the lanes of a wavefront hold CONSECUTIVE entries, so a gathered hop's
  // 32 or so live lanes ask for one run of the partner row -- 4-8 cache lines where pairs of entries per lane (the shape
  // of the chain kernel: 16-byte loads) spread them over 16-24, and the texture path was 90 % busy with those
  // (profiles/r05_kagome_real_counters.txt)
#define SC3R_ENT(i) (tsub + (i) * NTS)

  SC3_PRIO_MEM();
  uint32_t lowp[PPT];                                   // the Lo patterns of a pair's entries, 16 bits each
#define SC3R_PAT(i) ((lowp[(i) >> 1] >> (((i) & 1) * 16)) & 0xffffu)
  double xv[EPT];
  // the row's own pieces of x, y and the cached diagonal: uniform bases, one 32-bit offset per entry serves all three
  const double *__restrict__ xrow = x + base;
  double *__restrict__ yrow = y + lbase;
  const double *__restrict__ drow = DIAGM == 1 ? O.diag + lbase : nullptr;
  const auto pat = SC3_CP(uint16_t, S.lo_pat) + S.lo_off[kl];
#pragma unroll
  for (int i = 0; i < PPT; ++i) lowp[i] = 0;
  // (loads without a branch each: an entry behind the row's end reads the row's last one and drops it)
  const int rlast = nrows > 0 ? nrows - 1 : 0;
  if (has_row) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int r = SC3R_ENT(i), rc = min(r, rlast);
      const double xl = xrow[rc];
      const uint32_t pl = pat[rc];
      xv[i] = r < nrows ? xl : 0.0;
      lowp[i >> 1] |= (r < nrows ? pl : 0u) << ((i & 1) * 16);
    }
  } else {
#pragma unroll
    for (int i = 0; i < EPT; ++i) xv[i] = 0.0;
  }
  HopEval g{0, 0, 0.0, 0.0, 0, 0u, 0};
  if (lane < O.ngatA) {
    const Sc3Hop h = O.gatA[lane];
    const uint32_t T2 = T ^ h.mT, W2 = W ^ h.mW;
    const int nd = h.half - __popc(T & h.mT) - __popc(W & h.mW);
    const int nlo = __popc(h.mLo);
    const int kr2 = S.k - __popc(T2), cw2 = __popc(W2), kl2 = kr2 - cw2;
    if (nd >= 0 && nd <= nlo && kl2 >= 0 && kl2 <= A && cw2 <= w) {
      const int64_t tb2 = S.ibase[T2];
      if (tb2 >= 0) {
        g.act = 1;
        g.delta = tb2 + S.icoff[kr2 * (w + 1) + cw2] + (int64_t)S.w_rank[W2] * S.pitch[kl2] - base;
        g.need = nd;
        g.xm = h.mLo;
        g.c0 = h.up_re;
#if DNM_SC3G_PROBE_STAGE
        g.rev = S.nl[kl2];
#endif
      }
    }
  }
  uint64_t hb = has_row ? __ballot(g.act) : 0ull;
#pragma unroll
  for (int i = 0; i < ((1 << H) + NT - 1) / NT; ++i) {
    const int tt = threadIdx.x + i * NT;
    if (tt < (1 << H)) rka[tt] = (uint16_t)(tka[i] << 3);         // ranks times 8: byte offsets of real entries
  }
#pragma unroll
  for (int i = 0; i < ((1 << HB) * (H + 1) + NT - 1) / NT; ++i) {
    const int tt = threadIdx.x + i * NT;
    if (tt < (1 << HB) * (H + 1)) rkb[tt] = (uint16_t)(tkb[i] << 3);
  }
  const unsigned char *rkab = reinterpret_cast<const unsigned char *>(rka), *rkbb = reinterpret_cast<const unsigned char *>(rkb);
  // byte offset of the entry of a flipped pattern p2 (as in sc3g_lo_pass; no registers here to keep an entry's own offsets)
  auto rank8 = [&](uint32_t p2) -> uint32_t {
    const uint32_t t = (p2 & HM) << 1;
    return (uint32_t)*reinterpret_cast<const uint16_t *>(rkab + t) +
           (uint32_t)*reinterpret_cast<const uint16_t *>(rkbb + (((p2 >> H) << (HS + 1)) | ((uint32_t)__popc(t) << 1)));
  };
  if (DIAGM == 2 && tsub < 64) {
    const uint64_t hi = ((uint64_t)T << w) | W;
    double v0 = 0.0, vm[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t0 = 0; t0 < O.ndt; t0 += 64) {
      const int t = t0 + lane;
      if (t < O.ndt) {
        const uint64_t sg = SC3_CP(uint64_t, O.dt_sign)[t];
        const double c = flip(SC3_CP(double, O.dt_coef)[t], (uint32_t)__popcll(hi & sg & 0x1fffffffffffffffull) & 1u);
        const int gq = (int)(sg >> 61);
        if (gq == 0) v0 += c;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (gq == j + 1) vm[j] += c;
      }
    }
    v0 = wave_sum(v0);
    if (lane == 0) dsh[5 * sub] = v0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < O.ngroups) {
        const double s = wave_sum(vm[j]);
        if (lane == 0) dsh[5 * sub + j + 1] = s;
      }
  }
  double acc[EPT];
  // the zero entry behind the row's entries, where the partner table points for hops that do not act: the padding of x
  // holds zeros by the layout's contract, but the tile does not rely on it; a row without padding gets the entry here
  if (O.ptab && has_row && tsub == 0) xs[nrows] = 0.0;
#pragma unroll
  for (int i = 0; i < EPT; ++i) acc[i] = 0.0;
  if (has_row) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int r = SC3R_ENT(i), rc = min(r, rlast);
      if (r < nrows) xs[r] = xv[i];
      // the accumulators start from what the window pass wrote (no registers hold y to the end); xv is zero behind the row
      if (ACC) {
        const double yl = __builtin_nontemporal_load(yrow + rc);
        acc[i] = r < nrows ? yl : 0.0;
      }
      if (DIAGM == 1) acc[i] = fma(__builtin_nontemporal_load(drow + rc), xv[i], acc[i]);
    }
  }
  __syncthreads();
  while (hb) {
    const int m = __ffsll((long long)hb) - 1;
    hb &= hb - 1;
    const double *__restrict__ pp = x + (base + rl_i64(g.delta, m));
    const double cr = rl_f64(g.c0, m);
    const int nd = rl_i32(g.need, m);
    const uint32_t xm = (uint32_t)rl_i32((int)g.xm, m);
    double v[EPT];
#if DNM_SC3G_PROBE_STAGE
    // TIMING PROBE (wrong numbers): what a gathered hop would cost with its partner row STAGED in LDS -- the row read by
    // coalesced 16-byte loads (the data goes nowhere), one LDS read per live entry (from the row's own tile)
    {
      const int nl2 = rl_i32(g.rev, m);
      double sink = 0.0;
#pragma unroll
      for (int c = 0; c < EPT / 2; ++c) {
        const int idx = 2 * (tsub + c * NTS);
        if (idx < nl2) {
          const d2v q = *reinterpret_cast<const d2v *>(pp + idx);
          sink += q.x + q.y;
        }
      }
      acc[0] = fma(sink, 1e-300, acc[0]);
      const uint32_t slotmask = (uint32_t)(((NT * EPT) >> logm) * 8 - 8);
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int r = SC3R_ENT(i);
        const uint32_t pt = SC3R_PAT(i);
        v[i] = 0.0;
        if (r < nrows && __popc(pt & xm) == nd)
          v[i] = *reinterpret_cast<const double *>(reinterpret_cast<const unsigned char *>(xs) + (rank8(pt ^ xm) & slotmask));
      }
    }
#else
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int r = SC3R_ENT(i);
      const uint32_t pt = SC3R_PAT(i);
      v[i] = 0.0;
      if (r < nrows && __popc(pt & xm) == nd)
        v[i] = *reinterpret_cast<const double *>(reinterpret_cast<const unsigned char *>(pp) + rank8(pt ^ xm));
    }
#endif
#pragma unroll
    for (int i = 0; i < EPT; ++i) acc[i] = fma(cr, v[i], acc[i]);
  }
  SC3_PRIO_LDS();
  if (DIAGM == 2) {
    const auto dl = SC3_CP(double, O.dlo) + S.lo_off[kl];
    const double dg0 = dsh[5 * sub];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int r = SC3R_ENT(i);
      if (r < nrows) {
        double dg = dl[r] + dg0;
        for (int j = 0; j < O.ngroups; ++j) dg += flip(dsh[5 * sub + j + 1], (uint32_t)__popc(SC3R_PAT(i) & O.glo[j]) & 1u);
        acc[i] = fma(dg, xs[r], acc[i]);
      }
    }
  }
  if (O.ptab) {
    // hops inside Lo by the partner table: the two table rows of a pair of entries (16-byte words, straight from the
    // L2: every row of the class reads the same ones), then per (entry, hop) one 16-bit extract, one LDS read, one FMA
    constexpr int NQ = SC3G_MAX_PTAB / 8;
    const int nq = O.nhp >> 3;
    const uint4 *__restrict__ trow = reinterpret_cast<const uint4 *>(O.ptab + (size_t)O.ptab_row[kl] * (size_t)O.nhp);
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(xs);
    const unsigned char *__restrict__ tb = reinterpret_cast<const unsigned char *>(trow);     // uniform base + 32-bit offsets
    const uint32_t rowb = (uint32_t)nq * 16u;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int r = SC3R_ENT(2 * j);
      // (two entries of the thread at a time; the second one's table row is the class's padding row -- every hop at the
      // zero entry -- where the row has ended)
      const int r2 = SC3R_ENT(2 * j + 1) < nrows ? SC3R_ENT(2 * j + 1) : nrows;
      if (r < nrows) {
        // four hops at a time, no branch among them (the table's last word is padded with hops of coefficient zero that
        // point at the zero entry): their LDS reads go out together
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
          if (c < nq) {
            const uint4 q0 = *reinterpret_cast<const uint4 *>(tb + ((uint32_t)r * rowb + 16u * c));
            const uint4 q1 = *reinterpret_cast<const uint4 *>(tb + ((uint32_t)r2 * rowb + 16u * c));
            const uint32_t d0[4] = {q0.x, q0.y, q0.z, q0.w}, d1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int g4 = 0; g4 < 2; ++g4) {
              __builtin_amdgcn_sched_barrier(0);
              double v0[4], v1[4];
#pragma unroll
              for (int hh = 0; hh < 4; ++hh) {
                const int h8 = 4 * g4 + hh;
                const uint32_t t0 = (h8 & 1) ? d0[h8 >> 1] >> 16 : d0[h8 >> 1] & 0xffffu;
                const uint32_t t1 = (h8 & 1) ? d1[h8 >> 1] >> 16 : d1[h8 >> 1] & 0xffffu;
                v0[hh] = *reinterpret_cast<const double *>(xb + t0);
                v1[hh] = *reinterpret_cast<const double *>(xb + t1);
              }
#pragma unroll
              for (int hh = 0; hh < 4; ++hh) {
                const double ure = SC3_CP(double, O.pcoef)[8 * c + 4 * g4 + hh];
                acc[2 * j] = fma(ure, v0[hh], acc[2 * j]);
                acc[2 * j + 1] = fma(ure, v1[hh], acc[2 * j + 1]);
              }
            }
          }
        }
      }
    }
  } else
  for (int hq = 0; hq < O.nldsA; ++hq) {
    const auto hp = SC3_CP(Sc3Hop, O.ldsA) + hq;
    const uint32_t m = hp->mLo;
    const int half = hp->half;
    const double ure = hp->up_re;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int r = SC3R_ENT(i);
      const uint32_t pt = SC3R_PAT(i);
      if (r < nrows && __popc(pt & m) == half)
        acc[i] = fma(ure, *reinterpret_cast<const double *>(reinterpret_cast<const unsigned char *>(xs) + rank8(pt ^ m)), acc[i]);
    }
  }
  double dr = 0.0, dn = 0.0;
  SC3_PRIO_MEM();
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int r = SC3R_ENT(i);
    if (r < p) {                                  // the padding of a row is written too (zeros)
      double a = 0.0;
      if (r < nrows) {
        a = acc[i];
        if (!ACC && C.zinit) {
          a = fma(-C.zscale, reinterpret_cast<const double *>(C.zinit)[lbase + r], a);
          if (C.zinit2) a = fma(C.z2re, reinterpret_cast<const double *>(C.zinit2)[lbase + r], a);
        }
        if (ACC && C.dot_out) {
          dr = fma(xs[r], a, dr);
          dn = fma(a, a, dn);
        }
      }
      __builtin_nontemporal_store(a, yrow + r);
    }
  }
#undef SC3R_ENT
#undef SC3R_PAT
  if (ACC && C.dot_out) {
    dr = wave_sum(dr); dn = wave_sum(dn);
    if (lane == 0) {
      red[3 * (threadIdx.x >> 6)] = dr;
      red[3 * (threadIdx.x >> 6) + 1] = 0.0;
      red[3 * (threadIdx.x >> 6) + 2] = dn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double sr = 0.0, sn = 0.0;
      for (int wv = 0; wv < NT / 64; ++wv) { sr += red[3 * wv]; sn += red[3 * wv + 2]; }
      C.dot_out[3 * (size_t)blockIdx.x] = sr;
      C.dot_out[3 * (size_t)blockIdx.x + 1] = 0.0;
      C.dot_out[3 * (size_t)blockIdx.x + 2] = sn;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// window pass: y (+)= (hops inside W from the LDS tile, hops between W and T and inside T gathered) x.  Workgroup =
// (T, cw, run of R = 16 << s columns), the tile all window patterns of the class x R columns, as sc3_win_pass.
// REALV: real vectors (DNM_MAT_REAL_PACKED): the kernel runs on the layout's halved position tables, an element being a
// pair of adjacent real entries that gets the same real coefficient -- it never looks inside a pair, except where
// XParity's flip-composed hops read their columns from the other end of the row: there the two entries of a pair come
// from two real columns, in reverse order.
template <int WB, int NT, bool SYM, bool ACC, bool REALV = false>
__global__ void __launch_bounds__(NT, sc3_win_waves(NT, (cbinom(WB, WB / 2) * 16 * 16 + 1023) / 1024 + 12))
sc3g_win_pass(const Sc3Tab S, const Sc3Op O, const uint32_t *__restrict__ perm, const Sc3Call C,
              const c128 *__restrict__ xw, c128 *__restrict__ y) {
  constexpr int MAXE = cbinom(WB, WB / 2) * 16;
  constexpr int RPT = (MAXE + NT - 1) / NT;
  constexpr uint32_t WM = (1u << WB) - 1u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint8_t wrk[1 << WB];               // rank of a window pattern inside its class
  c128 *xs = reinterpret_cast<c128 *>(smem);     // the tile [wr][column], then one zero row (wr = nwp)
  const uint32_t e = SC3_CP(uint32_t, perm)[blockIdx.x];
  if (e == 0xffffffffu) return;
  constexpr int NWR = ((1 << WB) + NT - 1) / NT;
  uint16_t twr[NWR];               // asked for first, stored with the tile
#pragma unroll
  for (int i = 0; i < NWR; ++i) {
    const int j = threadIdx.x + i * NT;
    twr[i] = j < (1 << WB) ? S.w_rank[j] : (uint16_t)0;
  }
  const int lane = threadIdx.x & 63;
  const uint32_t T = e >> 16;
  const int cw = (e >> 12) & 15, run = e & 0xfff;
  const int kr = S.k - __popc(T), kl = kr - cw;
  const int nwp = S.nw[cw], p = S.pitch[kl];
  const int sh = 4 + S.rs[cw];
  const int lr0 = run << sh;
  const int ncols = min(1 << sh, p - lr0);
  const int64_t tb = SC3_CP(int64_t, S.ibase)[T];
  const int64_t own = tb + SC3_CP(int64_t, S.icoff)[kr * (WB + 1) + cw];
  const int64_t cbase = own + lr0;
  const int64_t lcb = cbase - C.row0;
  const int nent = nwp << sh;
  const c128 *__restrict__ x = xw - C.win_start;

  SC3_PRIO_MEM();
  uint32_t wpat[RPT];
  int32_t off[RPT];          // offset of the entry from cbase, -1: not an entry
  c128 xv[RPT];
  const auto pat = SC3_CP(uint16_t, S.w_pat) + S.w_off[cw];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int en = threadIdx.x + i * NT;
    wpat[i] = 0;
    off[i] = -1;
    xv[i] = make_double2(0.0, 0.0);
    if (en < nent) {
      const int wrr = en >> sh, j = en & ((1 << sh) - 1);
      wpat[i] = pat[wrr] | ((uint32_t)wrr << 16);
      if (j < ncols) {
        off[i] = wrr * p + j;
        xv[i] = x[cbase + off[i]];
      }
    }
  }
  // gathered hops, one per lane: both spins in T (a uniform offset), or spin i in W and spin j in T (the rows of the
  // partner class by rank, the columns as they are)
  // (XParity's flip-composed hops complement all of Lo: the partner column is the row's column counted from the
  // other end -- the complement reverses the order of the patterns)
  HopEval g{0, 0, 0.0, 0.0, 0, 0u, 0};
  if (lane < O.ngatB) {
    const Sc3Hop h = O.gatB[lane];
    const uint32_t T2 = T ^ h.mT;
    const bool flip_lo = h.mLo != 0u;
    const int nd = h.half - __popc(T & h.mT) - (flip_lo ? kl : 0);
    const int nwm = __popc(h.mW);
    const int kr2 = S.k - __popc(T2), cw2 = cw + nwm - 2 * nd, kl2 = kr2 - cw2;
    if (nd >= 0 && nd <= nwm && cw2 >= 0 && cw2 <= WB && kl2 == (flip_lo ? S.a - kl : kl)) {
      const int64_t tb2 = S.ibase[T2];
      if (tb2 >= 0) {
        g.act = 1;
        g.delta = tb2 + S.icoff[kr2 * (WB + 1) + cw2] - own;
        g.need = nd;
        g.xm = h.mW;
        g.rev = flip_lo ? 1 : 0;
        const bool up = h.dfield == 3 || (h.dfield == 1 ? nd == 1 : ((T >> h.dbit) & 1u) != 0);
        g.c0 = up ? h.up_re : h.dn_re;
        g.c1 = up ? h.up_im : h.dn_im;
      }
    }
  }
  uint64_t hb = __ballot(g.act);
  for (int j = threadIdx.x; j < (1 << sh); j += NT) xs[nent + j] = make_double2(0.0, 0.0);       // the zero row
  // the partner rows of the hops inside W, behind it (Sc3Op::wnb: nldsB bytes per row of the class)
  uint8_t *wtab = reinterpret_cast<uint8_t *>(xs + nent + (1 << sh));
  const int nhl = O.nldsB;
  for (int j = threadIdx.x; j < nwp * nhl; j += NT) wtab[j] = O.wnb[(size_t)S.w_off[cw] * nhl + j];
#pragma unroll
  for (int i = 0; i < NWR; ++i) {
    const int j = threadIdx.x + i * NT;
    if (j < (1 << WB)) wrk[j] = (uint8_t)twr[i];
  }
  double accr[RPT], acci[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int en = threadIdx.x + i * NT;
    accr[i] = 0.0;
    acci[i] = 0.0;
    if (en < nent) {
      xs[en] = xv[i];
      if (!ACC && C.zinit && off[i] >= 0) {
        const c128 zv = C.zinit[lcb + off[i]];
        accr[i] = -C.zscale * zv.x;
        acci[i] = -C.zscale * zv.y;
        if (C.zinit2) {
          const c128 z2 = C.zinit2[lcb + off[i]];
          accr[i] = fma(C.z2re, z2.x, accr[i]);
          accr[i] = fma(-C.z2im, z2.y, accr[i]);
          acci[i] = fma(C.z2re, z2.y, acci[i]);
          acci[i] = fma(C.z2im, z2.x, acci[i]);
        }
      }
    }
  }
  __syncthreads();                 // tile, zero row and tables are in LDS (loads in flight are not waited for)
  while (hb) {
    const int m = __ffsll((long long)hb) - 1;
    hb &= hb - 1;
    const c128 *__restrict__ pp = x + (cbase + rl_i64(g.delta, m));
    const double cr = rl_f64(g.c0, m), ci = rl_f64(g.c1, m);
    const int nd = rl_i32(g.need, m);
    const uint32_t xm = (uint32_t)rl_i32((int)g.xm, m);
    const int rev = rl_i32(g.rev, m);
    c128 v[RPT];
    if (rev && REALV) {            // ... of REAL entries: element (row, j) holds the real columns 2 (lr0 + j) and the next
      const double *__restrict__ pr = reinterpret_cast<const double *>(pp - lr0);      // the partner class, in real entries
      const int nlr = S.nl[kl];
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        v[i] = make_double2(0.0, 0.0);
        const uint32_t wp = wpat[i] & WM;
        const int wrr = (int)(wpat[i] >> 16), c0 = 2 * (lr0 + off[i] - wrr * p);
        if (off[i] >= 0 && c0 < nlr && __popc(wp & xm) == nd) {
          const double *q = pr + (int64_t)wrk[wp ^ xm] * (2 * p) + (nlr - 1 - c0);
          v[i].x = q[0];
          if (c0 + 1 < nlr) v[i].y = q[-1];
        }
      }
    } else if (rev) {              // rows by rank, columns from the other end of the row
      const int last = S.nl[kl] - 1 - lr0;            // column of the partner of this run's first column, from cbase - lr0
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        v[i] = make_double2(0.0, 0.0);
        const uint32_t wp = wpat[i] & WM;
        const int wrr = (int)(wpat[i] >> 16), j = off[i] - wrr * p;
        if (off[i] >= 0 && j <= last && __popc(wp & xm) == nd) v[i] = pp[(int)wrk[wp ^ xm] * p + last - j - lr0];
      }
    } else if (xm == 0u) {         // both spins in T: every entry, the same offset
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        v[i] = make_double2(0.0, 0.0);
        if (off[i] >= 0) v[i] = pp[off[i]];
      }
    } else {
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        v[i] = make_double2(0.0, 0.0);
        const uint32_t wp = wpat[i] & WM;
        if (off[i] >= 0 && __popc(wp & xm) == nd) {
          const int wr2 = (int)wrk[wp ^ xm], wrr = (int)(wpat[i] >> 16);
          v[i] = pp[off[i] + (wr2 - wrr) * p];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      accr[i] = fma(cr, v[i].x, accr[i]);
      acci[i] = fma(cr, v[i].y, acci[i]);
      if (!SYM) {
        accr[i] = fma(-ci, v[i].y, accr[i]);
        acci[i] = fma(ci, v[i].x, acci[i]);
      }
    }
  }
  SC3_PRIO_LDS();
  // hops inside W: partner row from the operator's table; a row the hop does not act on points at the zero row
  {
    uint32_t col[RPT], trow[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int en = threadIdx.x + i * NT;
      col[i] = (uint32_t)en & ((1u << sh) - 1u);
      trow[i] = en < nent ? (wpat[i] >> 16) * (uint32_t)nhl : 0u;
    }
    for (int hq = 0; hq < nhl; ++hq) {
      const auto hp = SC3_CP(Sc3Hop, O.ldsB) + hq;
      const int dbit = hp->dbit;
      const double ure = hp->up_re, uim = hp->up_im, dre = hp->dn_re, dim_ = hp->dn_im;
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const int en = threadIdx.x + i * NT;
        const uint32_t pr = en < nent ? (uint32_t)wtab[trow[i] + hq] : (uint32_t)nwp;
        const c128 xp = xs[(pr << sh) + col[i]];
        if (SYM) {
          accr[i] = fma(ure, xp.x, accr[i]);
          acci[i] = fma(ure, xp.y, acci[i]);
        } else {
          const bool up = (wpat[i] >> dbit) & 1u;
          const double cre = up ? ure : dre, cim = up ? uim : dim_;
          accr[i] = fma(cre, xp.x, accr[i]);
          acci[i] = fma(cre, xp.y, acci[i]);
          accr[i] = fma(-cim, xp.y, accr[i]);
          acci[i] = fma(cim, xp.x, acci[i]);
        }
      }
    }
  }
  SC3_PRIO_MEM();
#pragma unroll
  for (int i = 0; i < RPT; ++i)
    if (off[i] >= 0) {
      double ar = accr[i], ai = acci[i];
      if (ACC) {
        const c128 yo = load_nt(y + lcb + off[i]);
        ar += yo.x;
        ai += yo.y;
      }
      store_nt(y + lcb + off[i], ar, ai);
    }
}

}  // namespace

// phase 0: window pass writes y, lo pass adds (one rank); phase 1: lo pass alone (writes y); phase 2: window pass (adds)
template <int A, int W, int NT, int NTW>
static int launch_graph_passes(const Sc3Mat &M, const Sc3Call &call, const double *cached_diag, const void *xw, void *y,
                               hipStream_t st, int phase) {
  const Sc3Tab &S = M.ly->dev;
  constexpr size_t ldsA = (size_t)sc3_lo_cap(A, NT) * 16;
  size_t ldsB = 0;
  for (int cw = 0; cw <= W; ++cw)
    ldsB = std::max(ldsB, (((size_t)M.ly->host.nw[cw] + 1) << (4 + M.ly->host.rs[cw] + 4)) +
                              (((size_t)M.ly->host.nw[cw] * (size_t)M.op.nldsB + 15) & ~(size_t)15));
  Sc3Op op = M.op;
  const int dm = M.diag_mode;
  if (dm == 1) op.diag = cached_diag;
  DNM_CHECK(dm != 1 || op.diag, "this operator needs its diagonal precomputed (dnm_mat_precompute_diagonal)");
  using kern_t = void (*)(const Sc3Tab, const Sc3Op, const uint32_t *, const Sc3Call, const c128 *, c128 *);
  const bool lo_first = phase != 0;
  kern_t kB = nullptr, kA = nullptr;
  if (lo_first) kB = M.sym ? sc3g_win_pass<W, NTW, true, true> : sc3g_win_pass<W, NTW, false, true>;
  else kB = M.sym ? sc3g_win_pass<W, NTW, true, false> : sc3g_win_pass<W, NTW, false, false>;
#define DNM_LO(DM_, SY_) (lo_first ? (kern_t)sc3g_lo_pass<A, NT, DM_, SY_, false> : (kern_t)sc3g_lo_pass<A, NT, DM_, SY_, true>)
  switch (dm * 2 + (M.sym ? 1 : 0)) {
    case 0: kA = DNM_LO(0, false); break;
    case 1: kA = DNM_LO(0, true); break;
    case 2: kA = DNM_LO(1, false); break;
    case 3: kA = DNM_LO(1, true); break;
    case 4: kA = DNM_LO(2, false); break;
    default: kA = DNM_LO(2, true); break;
  }
#undef DNM_LO
  static std::map<const void *, size_t> attr_done;
  for (auto kp : {std::make_pair((const void *)kA, ldsA), std::make_pair((const void *)kB, ldsB)})
    if (attr_done[kp.first] < kp.second) {
      DNM_HIP(hipFuncSetAttribute(kp.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kp.second));
      attr_done[kp.first] = kp.second;
    }
  Sc3Call first = call, second = call;
  first.dot_out = nullptr;
  second.zinit = nullptr;
  second.zinit2 = nullptr;
  if (M.real) {
    // real vectors: the window pass is the complex kernel on the halved tables (pairs of entries as elements; its REALV
    // instance takes the pairs apart where XParity's flip-composed hops read columns backwards), the lo pass its own
    // kernel on doubles
    DNM_CHECK(M.sym, "internal: real vectors need a real operator");
    using kern_r = void (*)(const Sc3Tab, const Sc3Op, const uint32_t *, const Sc3Call, const double *, double *);
    kern_r kR = nullptr;
    constexpr int NTR = sc3r_threads(NT), PPR = sc3r_pairs(A, NT);
#define DNM_LOR(DM_) (lo_first ? (kern_r)sc3g_lo_pass_r<A, NTR, PPR, DM_, false> : (kern_r)sc3g_lo_pass_r<A, NTR, PPR, DM_, true>)
    kR = dm == 0 ? DNM_LOR(0) : (dm == 1 ? DNM_LOR(1) : DNM_LOR(2));
#undef DNM_LOR
    constexpr size_t ldsR = (size_t)sc3_lo_cap_r(A, NT) * 8;
    if (attr_done[(const void *)kR] < ldsR) {
      DNM_HIP(hipFuncSetAttribute((const void *)kR, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsR));
      attr_done[(const void *)kR] = ldsR;
    }
    Sc3Call firstw = phase == 0 ? first : second;
    firstw.row0 /= 2;
    firstw.win_start /= 2;
    kern_t kBr = lo_first ? (kern_t)sc3g_win_pass<W, NTW, true, true, true> : (kern_t)sc3g_win_pass<W, NTW, true, false, true>;
    if (attr_done[(const void *)kBr] < ldsB) {
      DNM_HIP(hipFuncSetAttribute((const void *)kBr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
      attr_done[(const void *)kBr] = ldsB;
    }
    if (phase == 0 || phase == 2)
      hipLaunchKernelGGL(kBr, dim3((unsigned)M.permB.size()), dim3(NTW), ldsB, st, M.ly->dev_h, op, (const uint32_t *)M.d_permB,
                         firstw, (const c128 *)xw, (c128 *)y);
    if (phase == 0 || phase == 1)
      hipLaunchKernelGGL(kR, dim3((unsigned)(M.permA.size() / 8)), dim3(NTR), ldsR, st, S, op, (const uint32_t *)M.d_permA,
                         phase == 0 ? second : first, (const double *)xw, (double *)y);
    DNM_HIP(hipGetLastError());
    return 0;
  }
  if (phase == 0 || phase == 2)
    hipLaunchKernelGGL(kB, dim3((unsigned)M.permB.size()), dim3(NTW), ldsB, st, S, op, (const uint32_t *)M.d_permB,
                       phase == 0 ? first : second, (const c128 *)xw, (c128 *)y);
  if (phase == 0 || phase == 1)
    hipLaunchKernelGGL(kA, dim3((unsigned)(M.permA.size() / 8)), dim3(NT), ldsA, st, S, op, (const uint32_t *)M.d_permA,
                       phase == 0 ? second : first, (const c128 *)xw, (c128 *)y);
  DNM_HIP(hipGetLastError());
  return 0;
}

int launch_sc3g(const Sc3Mat &M, const Sc3Call &call, const double *cached_diag, const void *xw, void *y, hipStream_t st,
                int phase) {
  if (M.ly->host.a == 14) return launch_graph_passes<14, 10, 1024, 512>(M, call, cached_diag, xw, y, st, phase);
  return launch_graph_passes<6, 4, 256, 64>(M, call, cached_diag, xw, y, st, phase);
}

}  // namespace dnm
