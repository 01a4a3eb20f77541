// SpinConserve(L,k) x SpinConserve(L,k) multiply in the three-field internal layout (sc3.h): two tiled passes for
// nearest-neighbour chains, a row kernel for any other operator, and the layout's vector utilities.
// Semantics: MatMult_CPU_General (src/dynamite/_backend/bpetsc_template_2.c:371-412) with the index maps of
// bsubspace_impl.h:187-245; the cached-diagonal variant follows bpetsc_template_1.c:186-199.
#include "sc3.h"

#include <algorithm>
#include <map>
#include <memory>
#include <mutex>

#include "dnm_common.h"
#include "kernels.h"
#include "philox.h"
#include "sc3_dev.h"

namespace dnm {

namespace {

// ---------------------------------------------------------------------------------------------------------
// lo pass (the second, accumulating pass): y += (bonds inside Lo, the Lo/W boundary, whatever else bondsA names
// and the diagonal) x.  A workgroup takes 2^m rows (T, W), m = 0..3 (round 4): a row has C(a, kl) states -- 3432 at
// kl = 7, 2002 at kl = 5, 364 at kl = 3 -- and one row per 1024-thread workgroup left 44 % of the lanes idle at
// SpinConserve(32,16).  The workgroup splits into 2^m sub-groups of NT >> m threads (whole wavefronts), each with its
// own row and its own slice of the LDS tile; everything that is uniform per row is uniform per wavefront, as before.
// perm holds 8 entries per workgroup: entry j = row of sub-group j | m << 30, bit 29 set = no row (SC3_NOROW).
//   DIAGM 0: no diagonal; 1: cached (internal order, 8 B/row); 2: on the fly -- terms that see Lo only from a table
//   over (kl, lr) (L2-resident), terms that see (T, W) only as one number per row, terms that see both as at most
//   four (Lo sign mask, per-row coefficient) pairs.
//   SYM: every bond coefficient is real and the same in both directions (Heisenberg, XXZ, XX chains).
//   ACC: true = the second pass (y += ...); false = it runs first and writes y, starting from the solver's start
//   vectors -- the order of a partitioned multiply, where this pass needs nothing from other ranks (a rank owns whole
//   T blocks) and runs while the window of x is still on the links.
template <int A, int NT, int DIAGM, bool SYM, bool ACC>
__global__ void __launch_bounds__(NT, sc3_win_waves(NT, (sc3_lo_cap(A, NT) * 16 + 1023) / 1024 + 1))
sc3_lo_pass(const Sc3Tab S, const Sc3Op O, const uint32_t *__restrict__ perm, const Sc3Call C,
            const c128 *__restrict__ xw, c128 *__restrict__ y) {
  constexpr int MAXROWS = cbinom(A, A / 2);
  constexpr int RPT = (MAXROWS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int32_t cl[A * (A + 1)];
  __shared__ double red[3 * (NT / 64)];
  __shared__ double dsh[5 * 8];
  const uint32_t e0 = SC3_CP(uint32_t, perm)[8 * (size_t)blockIdx.x];
  if (e0 == 0xffffffffu) return;                        // padding of the dispatch order: a workgroup without rows
  const int lane = threadIdx.x & 63;
  const int w = S.w;
  // sub-group of this wavefront: NTS threads, RPT entries each, its own slice of the tile
  const int logm = (int)(e0 >> 30);
  const int NTS = NT >> logm;
  const int sub = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >> (ilog2c(NT) - 6 - logm);
  const int tsub = (int)threadIdx.x & (NTS - 1);
  const uint32_t e = SC3_CP(uint32_t, perm)[8 * (size_t)blockIdx.x + sub];
  const bool has_row = !(e & SC3_NOROW);
  c128 *xs = reinterpret_cast<c128 *>(smem) + (size_t)sub * ((NT * RPT) >> logm);
  const RowId R = decode_row(has_row ? (e & (SC3_NOROW - 1u)) : (e0 & (SC3_NOROW - 1u)), S);
  const uint32_t T = R.T, W = R.W;
  const int cw = R.cw, kr = R.kr, kl = R.kl, nrows = has_row ? R.nrows : 0, p = has_row ? R.pitch : 0;
  const int64_t tb = R.tb, base = R.base;
  const c128 *__restrict__ x = xw - C.win_start;
  const int64_t lbase = base - C.row0;                 // position of the row in this rank's vectors

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
  // bonds outside Lo, one per lane: l = 0 the Lo/W boundary, 1..w-1 inside W, w the W/T boundary, above inside T.
  // Each couples this row to one other row at a uniform offset (the boundary bond: a contiguous part of it).
  int act = 0, r0 = 0, r1 = nrows;
  int64_t delta = 0;
  double c0 = 0.0, c1 = 0.0;
  {
    const int b = A - 1 + lane;
    if (b < S.L - 1 && ((O.bondsA >> b) & 1ull)) {
      bool up = false;
      if (lane == 0) {
        const int cut = SC3_CP(int32_t, S.cbin)[(A - 1) * 17 + kl];                   // rows below: top bit of Lo clear
        if (W & 1u) {                                                // the one comes down into Lo
          if (cut > 0) {
            act = 1; r0 = 0; r1 = cut; up = false;
            delta = tb + SC3_CP(int64_t, S.icoff)[kr * (w + 1) + cw - 1] + (int64_t)SC3_CP(uint16_t, S.w_rank)[W & ~1u] * S.pitch[kl + 1] +
                    SC3_CP(int32_t, S.cbin)[(A - 1) * 17 + kl + 1] - base;
          }
        } else if (cut < nrows) {                                    // the one goes up into W
          act = 1; r0 = cut; r1 = nrows; up = true;
          delta = tb + SC3_CP(int64_t, S.icoff)[kr * (w + 1) + cw + 1] + (int64_t)SC3_CP(uint16_t, S.w_rank)[W | 1u] * S.pitch[kl - 1] - cut - base;
        }
      } else if (lane < w) {
        const int bw = lane - 1;
        const uint32_t pair = (W >> bw) & 3u;
        if (pair == 1u || pair == 2u) {
          act = 1; up = pair == 1u;
          delta = ((int64_t)SC3_CP(uint16_t, S.w_rank)[W ^ (3u << bw)] - (int64_t)SC3_CP(uint16_t, S.w_rank)[W]) * p;
        }
      } else if (lane == w) {
        const uint32_t pair = ((W >> (w - 1)) & 1u) | ((T & 1u) << 1);
        if (pair == 1u) {
          act = 1; up = true;
          delta = SC3_CP(int64_t, S.ibase)[T | 1u] + SC3_CP(int64_t, S.icoff)[(kr - 1) * (w + 1) + cw - 1] +
                  (int64_t)SC3_CP(uint16_t, S.w_rank)[W & ~(1u << (w - 1))] * p - base;
        } else if (pair == 2u) {
          act = 1; up = false;
          delta = SC3_CP(int64_t, S.ibase)[T & ~1u] + SC3_CP(int64_t, S.icoff)[(kr + 1) * (w + 1) + cw + 1] +
                  (int64_t)SC3_CP(uint16_t, S.w_rank)[W | (1u << (w - 1))] * p - base;
        }
      } else {
        const int bt = lane - w - 1;
        const uint32_t pair = (T >> bt) & 3u;
        if (pair == 1u || pair == 2u) {
          act = 1; up = pair == 1u;
          delta = SC3_CP(int64_t, S.ibase)[T ^ (3u << bt)] - tb;
        }
      }
      if (act) {
        c0 = SC3_CP(double, O.bond)[4 * b + (up ? 0 : 2)];
        c1 = SC3_CP(double, O.bond)[4 * b + (up ? 1 : 3)];
      }
    }
  }
  uint64_t hb = has_row ? __ballot(act) : 0ull;

  for (int tt = threadIdx.x; tt < A * (A + 1); tt += NT) {
    const int lo = tt / (A + 1), o = tt % (A + 1);
    cl[tt] = SC3_CP(int32_t, S.cbin)[lo * 17 + o];
  }
  // on-the-fly diagonal: what the row (T, W) contributes -- group 0 to every state of the row, groups 1..4 with the
  // sign of a Lo pattern.  The first wavefront evaluates the terms, one per lane, and leaves the sums in LDS; they
  // are applied after the barrier the tile needs anyway.
  if (DIAGM == 2 && tsub < 64) {
    const uint64_t hi = ((uint64_t)T << w) | W;
    double v0 = 0.0, vm[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t0 = 0; t0 < O.ndt; t0 += 64) {
      const int t = t0 + lane;
      if (t < O.ndt) {
        const uint64_t sg = SC3_CP(uint64_t, O.dt_sign)[t];                       // bits 61..63: the group
        const double c = flip(SC3_CP(double, O.dt_coef)[t], (uint32_t)__popcll(hi & sg & 0x1fffffffffffffffull) & 1u);
        const int g = (int)(sg >> 61);
        if (g == 0) v0 += c;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (g == j + 1) vm[j] += c;
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
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = tsub + i * NTS;
    accr[i] = 0.0;
    acci[i] = 0.0;
    if (r < nrows) {
      xs[r] = xv[i];
      if (DIAGM == 1) {
        const double dg = __builtin_nontemporal_load(O.diag + lbase + r);
        accr[i] = dg * xv[i].x;
        acci[i] = dg * xv[i].y;
      }
    }
  }
  while (hb) {
    const int m = __ffsll((long long)hb) - 1;
    hb &= hb - 1;
    const c128 *__restrict__ pp = x + (base + rl_i64(delta, m));
    const double cr = rl_f64(c0, m), ci = rl_f64(c1, m);
    const int q0 = rl_i32(r0, m), q1 = rl_i32(r1, m);
    c128 v[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = tsub + i * NTS;
      v[i] = make_double2(0.0, 0.0);
      if (r >= q0 && r < q1) v[i] = pp[r];
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
  // the accumulating pass asks for its y now: it arrives while the bonds are served from LDS (the x values of the
  // entries are read back from the tile below, so their registers are free for it)
  c128 yv[RPT];
  if (ACC) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = tsub + i * NTS;
      yv[i] = make_double2(0.0, 0.0);
      if (r < nrows) yv[i] = load_nt(y + lbase + r);
    }
  }
  double dlv[RPT];
  if (DIAGM == 2) {          // the Lo-only part of the diagonal (L2-resident table), asked for before the barrier as well
    const auto dl = SC3_CP(double, O.dlo) + S.lo_off[kl];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = tsub + i * NTS;
      dlv[i] = r < nrows ? dl[r] : 0.0;
    }
  }
  __syncthreads();
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
  // bonds inside Lo.  (A partner table per (entry, bond) -- what the window pass uses, Sc3Tab::w_nb -- was measured here
  // too: 32 % fewer vector instructions and 0.3-0.9 ms MORE time, its 24 B per entry come from the L2 and every bond
  // then reads LDS, profiles/r03_exp11_sc3_tables.txt: this pass is not bound by its instructions.)
  for (int lo = 0; lo < A - 1; ++lo) {
    if (!((O.present >> lo) & 1ull)) continue;
    const double ure = SC3_CP(double, O.bond)[4 * lo], uim = SC3_CP(double, O.bond)[4 * lo + 1], dre = SC3_CP(double, O.bond)[4 * lo + 2], dim_ = SC3_CP(double, O.bond)[4 * lo + 3];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = tsub + i * NTS;
      const uint32_t pair = (lowb[i] >> lo) & 3u;
      if (r < nrows && (pair == 1u || pair == 2u)) {
        const bool up = pair == 1u;
        const int ord0 = __popc(lowb[i] & ((1u << lo) - 1u));
        const int d = cl[lo * (A + 1) + ord0];
        const c128 xp = xs[up ? r + d : r - d];
        if (SYM) {
          accr[i] = fma(ure, xp.x, accr[i]);
          acci[i] = fma(ure, xp.y, acci[i]);
        } else {
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
        if (ACC) {
          ar += yv[i].x;
          ai += yv[i].y;
        } else if (C.zinit) {                     // y = A x - b z (+ c z2): the first pass carries the start vectors
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
        if (ACC && C.dot_out) {                   // <x, y> and |y|^2 of the finished rows (the row of x is in LDS)
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
// lo pass on REAL vectors (DNM_MAT_REAL_PACKED on a SpinConserve pair, round 4): x and y are arrays of doubles in the
// same positions of the layout.  A thread owns PAIRS of adjacent row entries (2q, 2q + 1) -- rows start at multiples of
// 8 entries, so a pair is one aligned 16-byte access -- and twice as many entries as the complex pass (the same 64 KB
// tile holds 8192 doubles: two rows of 3432 states per workgroup, four of 2002 ...); the tile in LDS is an array of
// doubles, so the bonds inside Lo (partner r +- d, d of either parity) read 8 bytes.  Real operators have equal `up`
// and `dn` elements (Hermitian and real), so there is only the SYM form.  Everything per row -- sub-groups, the bond
// table in the lanes, the on-the-fly diagonal -- is as in sc3_lo_pass.
template <int A, int NT, int PPT, int DIAGM, bool ACC>
__global__ void __launch_bounds__(NT, sc3_win_waves(NT, (NT * 2 * PPT * 8 + 1023) / 1024 + 1))
sc3_lo_pass_r(const Sc3Tab S, const Sc3Op O, const uint32_t *__restrict__ perm, const Sc3Call C,
              const double *__restrict__ xw, double *__restrict__ y) {
  constexpr int MAXROWS = cbinom(A, A / 2);
  constexpr int EPT = 2 * PPT;                          // entries per thread (PPT pairs)
  static_assert(NT * EPT >= MAXROWS, "the longest row does not fit the workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int32_t cl[A * (A + 1)];
  __shared__ double red[3 * (NT / 64)];
  __shared__ double dsh[5 * 8];
  const uint32_t e0 = SC3_CP(uint32_t, perm)[8 * (size_t)blockIdx.x];
  if (e0 == 0xffffffffu) return;
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
  const int cw = R.cw, kr = R.kr, kl = R.kl, nrows = has_row ? R.nrows : 0, p = has_row ? R.pitch : 0;
  const int64_t tb = R.tb, base = R.base;
  const double *__restrict__ x = xw - C.win_start;
  const int64_t lbase = base - C.row0;
  // entry i of this thread: pair i >> 1, lane i & 1
#define SC3R_ENT(i) (2 * (tsub + ((i) >> 1) * NTS) + ((i) & 1))
  // where entry r of the row sits in the LDS slice.  The lanes of a wavefront read entries two apart (their own lane
  // of consecutive pairs, shifted by the bond's offset): two lanes per bank.  -DDNM_SC3R_DEINT=1 stores even entries
  // in the first half of the slice and odd ones in the second, which removes the conflicts -- and measured SLOWER
  // (5.11 against 4.45 ms at SpinConserve(32,16): the index arithmetic and 12-16 B/lane of spills cost more than the
  // conflicts; gpurun_out/r04_s16) -- so the interleaved form is the default
#ifndef DNM_SC3R_DEINT
#define DNM_SC3R_DEINT 0
#endif
#if DNM_SC3R_DEINT
  const int half = (NT * EPT >> logm) >> 1;
#define SC3R_LDS(r) (((r) >> 1) + (((r) & 1) ? half : 0))
#else
#define SC3R_LDS(r) (r)
#endif

  SC3_PRIO_MEM();
  uint32_t lowp[PPT];                                   // the Lo patterns of a pair's entries, 16 bits each
#define SC3R_PAT(i) ((lowp[(i) >> 1] >> (((i) & 1) * 16)) & 0xffffu)
  double xv[EPT];
  const auto pat = SC3_CP(uint16_t, S.lo_pat) + S.lo_off[kl];
#pragma unroll
  for (int i = 0; i < EPT; i += 2) {
    const int r = SC3R_ENT(i);
    lowp[i >> 1] = 0;
    xv[i] = xv[i + 1] = 0.0;
    if (r < p) {                                        // (the padding of a row holds zeros)
      const d2v v = *reinterpret_cast<const d2v *>(x + base + r);
      xv[i] = v.x;
      xv[i + 1] = v.y;
      if (r < nrows) lowp[i >> 1] = pat[r];
      if (r + 1 < nrows) lowp[i >> 1] |= (uint32_t)pat[r + 1] << 16;
    }
  }
  int act = 0, r0 = 0, r1 = nrows;
  int64_t delta = 0;
  double c0 = 0.0;
  {
    const int b = A - 1 + lane;
    if (b < S.L - 1 && ((O.bondsA >> b) & 1ull)) {
      bool up = false;
      if (lane == 0) {
        const int cut = SC3_CP(int32_t, S.cbin)[(A - 1) * 17 + kl];
        if (W & 1u) {
          if (cut > 0) {
            act = 1; r0 = 0; r1 = cut; up = false;
            delta = tb + SC3_CP(int64_t, S.icoff)[kr * (w + 1) + cw - 1] + (int64_t)SC3_CP(uint16_t, S.w_rank)[W & ~1u] * S.pitch[kl + 1] +
                    SC3_CP(int32_t, S.cbin)[(A - 1) * 17 + kl + 1] - base;
          }
        } else if (cut < nrows) {
          act = 1; r0 = cut; r1 = nrows; up = true;
          delta = tb + SC3_CP(int64_t, S.icoff)[kr * (w + 1) + cw + 1] + (int64_t)SC3_CP(uint16_t, S.w_rank)[W | 1u] * S.pitch[kl - 1] - cut - base;
        }
      } else if (lane < w) {
        const int bw = lane - 1;
        const uint32_t pair = (W >> bw) & 3u;
        if (pair == 1u || pair == 2u) {
          act = 1; up = pair == 1u;
          delta = ((int64_t)SC3_CP(uint16_t, S.w_rank)[W ^ (3u << bw)] - (int64_t)SC3_CP(uint16_t, S.w_rank)[W]) * p;
        }
      } else if (lane == w) {
        const uint32_t pair = ((W >> (w - 1)) & 1u) | ((T & 1u) << 1);
        if (pair == 1u) {
          act = 1; up = true;
          delta = SC3_CP(int64_t, S.ibase)[T | 1u] + SC3_CP(int64_t, S.icoff)[(kr - 1) * (w + 1) + cw - 1] +
                  (int64_t)SC3_CP(uint16_t, S.w_rank)[W & ~(1u << (w - 1))] * p - base;
        } else if (pair == 2u) {
          act = 1; up = false;
          delta = SC3_CP(int64_t, S.ibase)[T & ~1u] + SC3_CP(int64_t, S.icoff)[(kr + 1) * (w + 1) + cw + 1] +
                  (int64_t)SC3_CP(uint16_t, S.w_rank)[W | (1u << (w - 1))] * p - base;
        }
      } else {
        const int bt = lane - w - 1;
        const uint32_t pair = (T >> bt) & 3u;
        if (pair == 1u || pair == 2u) {
          act = 1; up = pair == 1u;
          delta = SC3_CP(int64_t, S.ibase)[T ^ (3u << bt)] - tb;
        }
      }
      if (act) c0 = SC3_CP(double, O.bond)[4 * b + (up ? 0 : 2)];
    }
  }
  uint64_t hb = has_row ? __ballot(act) : 0ull;

  for (int tt = threadIdx.x; tt < A * (A + 1); tt += NT) {
    const int lo = tt / (A + 1), o = tt % (A + 1);
    cl[tt] = SC3_CP(int32_t, S.cbin)[lo * 17 + o];
  }
  if (DIAGM == 2 && tsub < 64) {
    const uint64_t hi = ((uint64_t)T << w) | W;
    double v0 = 0.0, vm[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t0 = 0; t0 < O.ndt; t0 += 64) {
      const int t = t0 + lane;
      if (t < O.ndt) {
        const uint64_t sg = SC3_CP(uint64_t, O.dt_sign)[t];
        const double c = flip(SC3_CP(double, O.dt_coef)[t], (uint32_t)__popcll(hi & sg & 0x1fffffffffffffffull) & 1u);
        const int g = (int)(sg >> 61);
        if (g == 0) v0 += c;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (g == j + 1) vm[j] += c;
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
#pragma unroll
  for (int i = 0; i < EPT; i += 2) {
    const int r = SC3R_ENT(i);
    acc[i] = acc[i + 1] = 0.0;
    if (r < p) {
#if DNM_SC3R_DEINT
      xs[r >> 1] = xv[i];
      xs[half + (r >> 1)] = xv[i + 1];
#else
      *reinterpret_cast<d2v *>(xs + r) = d2v{xv[i], xv[i + 1]};
#endif
      if (DIAGM == 1) {
        const d2v dg = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(O.diag + lbase + r));
        acc[i] = dg.x * xv[i];
        acc[i + 1] = dg.y * xv[i + 1];
      }
    }
  }
  while (hb) {
    const int m = __ffsll((long long)hb) - 1;
    hb &= hb - 1;
    const int64_t dl_ = rl_i64(delta, m);
    const double *__restrict__ pp = x + (base + dl_);
    const double cr = rl_f64(c0, m);
    const int q0 = rl_i32(r0, m), q1 = rl_i32(r1, m);
    double v[EPT];
    if (!(dl_ & 1)) {                 // whole elements line up (every bond but, mostly, the Lo/W boundary)
#pragma unroll
      for (int i = 0; i < EPT; i += 2) {
        const int r = SC3R_ENT(i);
        v[i] = v[i + 1] = 0.0;
        if (r >= q0 && r + 1 < q1) {                // the whole pair lies in [q0, q1): one 16-byte load
          const d2v t = *reinterpret_cast<const d2v *>(pp + r);
          v[i] = t.x;
          v[i + 1] = t.y;
        } else {                                    // a pair that straddles an end of the range
          if (r >= q0 && r < q1) v[i] = pp[r];
          if (r + 1 >= q0 && r + 1 < q1) v[i + 1] = pp[r + 1];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int r = SC3R_ENT(i);
        v[i] = 0.0;
        if (r >= q0 && r < q1) v[i] = pp[r];
      }
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) acc[i] = fma(cr, v[i], acc[i]);
  }
  d2v yv[PPT];
  if (ACC) {
#pragma unroll
    for (int i = 0; i < EPT; i += 2) {
      const int r = SC3R_ENT(i);
      yv[i >> 1] = d2v{0.0, 0.0};
      if (r < p) yv[i >> 1] = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(y + lbase + r));
    }
  }
  __syncthreads();
  SC3_PRIO_LDS();
  if (DIAGM == 2) {
    // (the Lo-only part of the diagonal comes from its L2-resident table here, not before the barrier as in the
    // complex pass: eight entries per thread leave no registers to carry it across)
    const auto dl = SC3_CP(double, O.dlo) + S.lo_off[kl];
    const double dg0 = dsh[5 * sub];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int r = SC3R_ENT(i);
      if (r < nrows) {
        double dg = dl[r] + dg0;
        for (int j = 0; j < O.ngroups; ++j) dg += flip(dsh[5 * sub + j + 1], (uint32_t)__popc(SC3R_PAT(i) & O.glo[j]) & 1u);
        acc[i] = fma(dg, xs[SC3R_LDS(r)], acc[i]);
      }
    }
  }
  for (int lo = 0; lo < A - 1; ++lo) {
    if (!((O.present >> lo) & 1ull)) continue;
    const double ure = SC3_CP(double, O.bond)[4 * lo];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int r = SC3R_ENT(i);
      const uint32_t pt = SC3R_PAT(i);
      const uint32_t pair = (pt >> lo) & 3u;
      if (r < nrows && (pair == 1u || pair == 2u)) {
        const int ord0 = __popc(pt & ((1u << lo) - 1u));
        const int d = cl[lo * (A + 1) + ord0];
        const int rp = pair == 1u ? r + d : r - d;
        acc[i] = fma(ure, xs[SC3R_LDS(rp)], acc[i]);
      }
    }
  }
  double dr = 0.0, dn = 0.0;
  SC3_PRIO_MEM();
#pragma unroll
  for (int i = 0; i < EPT; i += 2) {
    const int r = SC3R_ENT(i);
    if (r < p) {                                  // the padding of a row is written too (zeros)
      double a2[2] = {acc[i], acc[i + 1]};
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (r + b < nrows) {
          if (ACC) a2[b] += b ? yv[i >> 1].y : yv[i >> 1].x;
          else if (C.zinit) {
            a2[b] = fma(-C.zscale, reinterpret_cast<const double *>(C.zinit)[lbase + r + b], a2[b]);
            if (C.zinit2) a2[b] = fma(C.z2re, reinterpret_cast<const double *>(C.zinit2)[lbase + r + b], a2[b]);
          }
          if (ACC && C.dot_out) {
            dr = fma(xs[SC3R_LDS(r + b)], a2[b], dr);
            dn = fma(a2[b], a2[b], dn);
          }
        } else {
          a2[b] = 0.0;
        }
      }
      __builtin_nontemporal_store(d2v{a2[0], a2[1]}, reinterpret_cast<d2v *>(y + lbase + r));
    }
  }
#undef SC3R_ENT
#undef SC3R_PAT
#undef SC3R_LDS
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
// window pass (the first pass: writes y): one workgroup per (T, cw, run of R = 16 << s columns): all window
// patterns of the class x R columns in LDS; the accumulators start from -zscale * zinit + z2 * zinit2 if given.
// (512 threads x 8 entries against 1024 x 4: 5.40 against 5.75 ms at SpinConserve(32,16), level on a rank of config 5,
// profiles/r03_exp12_sc3_win512.txt.  Two gathered bonds in flight at a time -- half the round trips of a workgroup's
// life, 12 of them on a rank of config 5 -- do not fit: the compiler needs 12-13 registers per entry where 8 are live,
// and the spills cost more than the round trips: 22 ms.)
template <int WB, int NT, bool SYM, bool ACC>
__global__ void __launch_bounds__(NT, sc3_win_waves(NT, (cbinom(WB, WB / 2) * 16 * 16 + 1023) / 1024 + 1))
sc3_win_pass(const Sc3Tab S, const Sc3Op O, const uint32_t *__restrict__ perm, const Sc3Call C,
             const c128 *__restrict__ xw, c128 *__restrict__ y) {
  constexpr int MAXE = cbinom(WB, WB / 2) * 16;
  constexpr int RPT = (MAXE + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c128 *xs = reinterpret_cast<c128 *>(smem);       // the tile [wr][column], then one zero row (wr = nwp)
  const uint32_t e = SC3_CP(uint32_t, perm)[blockIdx.x];
  if (e == 0xffffffffu) return;
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
  // gathered bonds, one per lane: l = 0 the W/T boundary, above inside T
  int act = 0, r0 = 0, r1 = nwp;
  int64_t delta = 0;
  double c0 = 0.0, c1 = 0.0;
  {
    const int b = S.a + WB - 1 + lane;
    if (b < S.L - 1 && ((O.bondsB >> b) & 1ull)) {
      bool up = false;
      if (lane == 0) {
        const int cut = SC3_CP(int32_t, S.cbin)[(WB - 1) * 17 + cw];                // rows below: top bit of W clear
        if (T & 1u) {                                              // the one comes down into W
          if (cut > 0) {
            act = 1; r0 = 0; r1 = cut; up = false;
            delta = SC3_CP(int64_t, S.ibase)[T & ~1u] + SC3_CP(int64_t, S.icoff)[(kr + 1) * (WB + 1) + cw + 1] +
                    (int64_t)SC3_CP(int32_t, S.cbin)[(WB - 1) * 17 + cw + 1] * p - own;
          }
        } else if (cut < nwp) {                                    // the one goes up into T
          act = 1; r0 = cut; r1 = nwp; up = true;
          delta = SC3_CP(int64_t, S.ibase)[T | 1u] + SC3_CP(int64_t, S.icoff)[(kr - 1) * (WB + 1) + cw - 1] - (int64_t)cut * p - own;
        }
      } else {
        const int bt = lane - 1;
        const uint32_t pair = (T >> bt) & 3u;
        if (pair == 1u || pair == 2u) {
          act = 1; up = pair == 1u;
          delta = SC3_CP(int64_t, S.ibase)[T ^ (3u << bt)] - tb;
        }
      }
      if (act) {
        c0 = SC3_CP(double, O.bond)[4 * b + (up ? 0 : 2)];
        c1 = SC3_CP(double, O.bond)[4 * b + (up ? 1 : 3)];
      }
    }
  }
  uint64_t hb = __ballot(act);
  for (int j = threadIdx.x; j < (1 << sh); j += NT) xs[nent + j] = make_double2(0.0, 0.0);       // the zero row
  // the class's partner table (Sc3Tab::w_nb, 16 bytes per row) behind it: read back from LDS after the barrier
  ulonglong2 *wtab = reinterpret_cast<ulonglong2 *>(xs + nent + (1 << sh));
  for (int j = threadIdx.x; j < nwp; j += NT)
    wtab[j] = reinterpret_cast<const ulonglong2 *>(S.w_nb + (size_t)2 * S.w_off[cw])[j];
  double accr[RPT], acci[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int en = threadIdx.x + i * NT;
    accr[i] = 0.0;
    acci[i] = 0.0;
    if (en < nent) {
      xs[en] = xv[i];
      if (!ACC && C.zinit && off[i] >= 0) {        // y = A x - b z (+ c z2): the start vectors open the accumulators
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
  while (hb) {
    const int m = __ffsll((long long)hb) - 1;
    hb &= hb - 1;
    const c128 *__restrict__ pp = x + (cbase + rl_i64(delta, m));
    const double cr = rl_f64(c0, m), ci = rl_f64(c1, m);
    const int q0 = rl_i32(r0, m), q1 = rl_i32(r1, m);
    c128 v[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int wrr = (int)(wpat[i] >> 16);
      v[i] = make_double2(0.0, 0.0);
      if (off[i] >= 0 && wrr >= q0 && wrr < q1) v[i] = pp[off[i]];
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
  __syncthreads();
  SC3_PRIO_LDS();
  // bonds inside W: the partner row of (row, bond) from the layout's table (Sc3Tab::w_nb) -- a row whose two spins are
  // equal points at the zero row behind the tile, so the loop has no branches
  {
    uint64_t t0[RPT], t1[RPT];
    uint32_t col[RPT];
    const uint64_t zr = (uint64_t)nwp * 0x0101010101010101ull;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int en = threadIdx.x + i * NT;
      const int wrr = (int)(wpat[i] >> 16);
      col[i] = (uint32_t)en & ((1u << sh) - 1u);
      t0[i] = zr;
      t1[i] = zr;
      if (en < nent) {
        const ulonglong2 tw = wtab[wrr];
        t0[i] = tw.x;
        if (WB - 1 > 8) t1[i] = tw.y;
      }
    }
#pragma unroll
    for (int lo = 0; lo < WB - 1; ++lo) {
      const int b = S.a + lo;
      if (!((O.present >> b) & 1ull)) continue;
      const double ure = SC3_CP(double, O.bond)[4 * b], uim = SC3_CP(double, O.bond)[4 * b + 1], dre = SC3_CP(double, O.bond)[4 * b + 2], dim_ = SC3_CP(double, O.bond)[4 * b + 3];
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const uint32_t pr = (uint32_t)((lo < 8 ? t0[i] : t1[i]) >> (8 * (lo & 7))) & 0xffu;
        const c128 xp = xs[(pr << sh) + col[i]];
        if (SYM) {
          accr[i] = fma(ure, xp.x, accr[i]);
          acci[i] = fma(ure, xp.y, acci[i]);
        } else {
          const bool up = (wpat[i] >> lo) & 1u;
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

// ---------------------------------------------------------------------------------------------------------
// Row kernel: any operator between SpinConserve(L,k) and itself in the internal layout.  One workgroup per row
// (T, W); a state's position is three table lookups (sc3_pos), so a column costs no unranking.  The matrix element
// follows bpetsc_template_2.c:396-405 (sign on the column state, TERM_REAL decides real / imaginary).
constexpr int SC3_ROW_NT = 256;
__global__ void __launch_bounds__(SC3_ROW_NT)
sc3_row_kernel(const Sc3Tab S, const DevMsc msc, const uint32_t *__restrict__ rows, const Sc3Call C,
               const double *__restrict__ diag, const c128 *__restrict__ xw, c128 *__restrict__ y) {
  const uint32_t e = rows[blockIdx.x];
  if (e == 0xffffffffu) return;
  const RowId R = decode_row(e, S);
  const c128 *__restrict__ x = xw - C.win_start;
  const int64_t lbase = R.base - C.row0;
  const uint64_t hi = (((uint64_t)R.T << S.w) | R.W) << S.a;
  const uint16_t *__restrict__ pat = S.lo_pat + S.lo_off[R.kl];
  for (int r = threadIdx.x; r < R.pitch; r += SC3_ROW_NT) {
    double accr = 0.0, acci = 0.0;
    if (r < R.nrows) {
      const uint64_t ket = hi | pat[r];
      if (C.zinit) {
        const c128 zv = C.zinit[lbase + r];
        accr = -C.zscale * zv.x;
        acci = -C.zscale * zv.y;
        if (C.zinit2) {
          const c128 z2 = C.zinit2[lbase + r];
          accr = fma(C.z2re, z2.x, accr);
          accr = fma(-C.z2im, z2.y, accr);
          acci = fma(C.z2re, z2.y, acci);
          acci = fma(C.z2im, z2.x, acci);
        }
      }
      int m0 = 0;
      if (diag) {
        const c128 xo = x[R.base + r];
        const double dg = diag[lbase + r];
        accr = fma(dg, xo.x, accr);
        acci = fma(dg, xo.y, acci);
        m0 = 1;
      }
      for (int m = m0; m < msc.nmasks; ++m) {
        const uint64_t mask = (uint64_t)msc.masks[m];
        const uint64_t bra = ket ^ mask;
        if (__popcll(bra) != S.k) continue;               // leaves the subspace: no such column
        double cre = 0.0, cim = 0.0;
        for (int64_t t = msc.mask_offsets[m]; t < msc.mask_offsets[m + 1]; ++t) {
          const uint64_t sg = (uint64_t)msc.signs[t];
          const double c = flip(msc.real_coeffs[t], (uint32_t)__popcll(bra & sg) & 1u);
          if (__popcll(mask & sg) & 1) cim += c; else cre += c;   // TERM_REAL (bpetsc_impl.h:34)
        }
        const c128 xv = x[sc3_pos(bra, S)];
        accr = fma(cre, xv.x, accr);
        acci = fma(cre, xv.y, acci);
        accr = fma(-cim, xv.y, accr);
        acci = fma(cim, xv.x, acci);
      }
    }
    store_nt(y + lbase + r, accr, acci);
  }
}

// ---------------------------------------------------------------------------------------------------------
// layout utilities: one workgroup per row (T, W)
template <typename V>
__global__ void __launch_bounds__(256)
sc3_copy_kernel(const Sc3Tab S, const uint32_t *__restrict__ rows, V *__restrict__ dst, const V *__restrict__ src,
                int to_internal, int64_t ioff, int64_t noff) {
  // ioff / noff: where the (partitioned) vectors start in the internal layout / the reference order
  const uint32_t e = rows[blockIdx.x];
  const RowId R = decode_row(e, S);
  const int64_t nat = S.nbase[R.T] + S.ncoff[(int64_t)R.kr * ((int64_t)1 << S.w) + R.W] - noff;
  const int64_t base = R.base - ioff;
  V zero{};
  for (int r = threadIdx.x; r < R.pitch; r += 256) {
    if (to_internal) dst[base + r] = r < R.nrows ? src[nat + r] : zero;
    else if (r < R.nrows) dst[nat + r] = src[base + r];
  }
}
__global__ void __launch_bounds__(64)
sc3_zero_pad_kernel(const Sc3Tab S, const uint32_t *__restrict__ rows, c128 *__restrict__ x, int64_t ioff) {
  const RowId R = decode_row(rows[blockIdx.x], S);
  for (int r = R.nrows + threadIdx.x; r < R.pitch; r += 64) x[R.base - ioff + r] = make_double2(0.0, 0.0);
}
__global__ void __launch_bounds__(256)
sc3_random_kernel(const Sc3Tab S, const uint32_t *__restrict__ rows, c128 *__restrict__ x, uint64_t seed, int64_t ioff) {
  const RowId R = decode_row(rows[blockIdx.x], S);
  const int64_t nat = S.nbase[R.T] + S.ncoff[(int64_t)R.kr * ((int64_t)1 << S.w) + R.W];
  for (int r = threadIdx.x; r < R.pitch; r += 256)
    x[R.base - ioff + r] = r < R.nrows ? philox_normal((uint64_t)(nat + r), seed) : make_double2(0.0, 0.0);
}
__global__ void __launch_bounds__(256)
sc3_random_real_kernel(const Sc3Tab S, const uint32_t *__restrict__ rows, double *__restrict__ x, uint64_t seed, int64_t ioff) {
  const RowId R = decode_row(rows[blockIdx.x], S);
  const int64_t nat = S.nbase[R.T] + S.ncoff[(int64_t)R.kr * ((int64_t)1 << S.w) + R.W];
  for (int r = threadIdx.x; r < R.pitch; r += 256)
    x[R.base - ioff + r] = r < R.nrows ? philox_normal((uint64_t)(nat + r), seed).x : 0.0;
}
__global__ void __launch_bounds__(256) sc3_unpack_real_kernel(c128 *__restrict__ dst, const double *__restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dst[i] = make_double2(src[i], 0.0);
}
// reference index -> internal position: unrank (bsubspace_impl.h:210-228), then the tables
__global__ void __launch_bounds__(256)
sc3_positions_kernel(const Sc3Tab S, int64_t n, const int64_t *__restrict__ idx, int64_t *__restrict__ pos, int64_t ioff,
                     int64_t noff) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t *__restrict__ nck = S.nck;
  const int ld = S.L + 1;
  int64_t id = idx[i] + noff;
  uint64_t st = 0;
  int k = S.k;
  for (int b = S.L; b > 0; --b) {
    const int64_t here = (k > b - 1) ? 0 : nck[(int64_t)k * ld + (b - 1)];
    st <<= 1;
    if (id >= here) { id -= here; --k; st |= 1; }
  }
  pos[i] = sc3_pos(st, S) - ioff;
}

// ---- the same utilities under a site relabelling (Sc3Perm; one rank, whole vectors) -----------------------
// reference index of a state given in the layout's labelling: back to the reference's spins, then the colex rank
// (bsubspace_impl.h:191-208)
__device__ __forceinline__ int64_t ref_index_of(uint64_t s_ref, const Sc3Tab &S) {
  int64_t idx = 0;
  int j = 0;
  const int ld = S.L + 1;
  while (s_ref) {
    const int n = __ffsll((long long)s_ref) - 1;
    ++j;
    if (j <= n) idx += S.nck[(int64_t)j * ld + n];
    s_ref &= s_ref - 1;
  }
  return idx;
}
template <typename V>
__global__ void __launch_bounds__(256)
sc3_copy_perm_kernel(const Sc3Tab S, const Sc3Perm P, const uint32_t *__restrict__ rows, V *__restrict__ dst,
                     const V *__restrict__ src, int to_internal) {
  const RowId R = decode_row(rows[blockIdx.x], S);
  const uint64_t hi = sc3_permute((((uint64_t)R.T << S.w) | R.W) << S.a, P.to_ref, S.L);
  const uint16_t *__restrict__ pat = S.lo_pat + S.lo_off[R.kl];
  V zero{};
  for (int r = threadIdx.x; r < R.pitch; r += 256) {
    if (r < R.nrows) {
      const int64_t nat = ref_index_of(hi | sc3_permute(pat[r], P.to_ref, S.a), S);
      if (to_internal) dst[R.base + r] = src[nat];
      else dst[nat] = src[R.base + r];
    } else if (to_internal) {
      dst[R.base + r] = zero;
    }
  }
}
template <bool REAL>
__global__ void __launch_bounds__(256)
sc3_random_perm_kernel(const Sc3Tab S, const Sc3Perm P, const uint32_t *__restrict__ rows, void *__restrict__ xv, uint64_t seed) {
  const RowId R = decode_row(rows[blockIdx.x], S);
  const uint64_t hi = sc3_permute((((uint64_t)R.T << S.w) | R.W) << S.a, P.to_ref, S.L);
  const uint16_t *__restrict__ pat = S.lo_pat + S.lo_off[R.kl];
  for (int r = threadIdx.x; r < R.pitch; r += 256) {
    c128 v = make_double2(0.0, 0.0);
    if (r < R.nrows) v = philox_normal((uint64_t)ref_index_of(hi | sc3_permute(pat[r], P.to_ref, S.a), S), seed);
    if (REAL) reinterpret_cast<double *>(xv)[R.base + r] = v.x;
    else reinterpret_cast<c128 *>(xv)[R.base + r] = v;
  }
}
__global__ void __launch_bounds__(256)
sc3_positions_perm_kernel(const Sc3Tab S, const Sc3Perm P, int64_t n, const int64_t *__restrict__ idx, int64_t *__restrict__ pos) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t *__restrict__ nck = S.nck;
  const int ld = S.L + 1;
  int64_t id = idx[i];
  uint64_t st = 0;
  int k = S.k;
  for (int b = S.L; b > 0; --b) {
    const int64_t here = (k > b - 1) ? 0 : nck[(int64_t)k * ld + (b - 1)];
    st <<= 1;
    if (id >= here) { id -= here; --k; st |= 1; }
  }
  pos[i] = sc3_pos(sc3_permute(st, P.to_int, S.L), S);
}

}  // namespace

// ===========================================================================================================
// host side
// ===========================================================================================================
static int64_t hbinom(int n, int k) {
  if (k < 0 || k > n) return 0;
  long double r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return (int64_t)llroundl(r);
}

bool sc3_valid(int L, int k, int a, int w) {
  const int t = L - a - w;
  return a >= 2 && a <= SC3_MAXA && w >= 2 && w <= SC3_MAXW && t >= 1 && t <= 15 && k >= 0 && k <= L;
}

Sc3Layout::~Sc3Layout() {
  for (void *p : {d_ibase, d_nbase, d_icoff, d_ncoff, d_lo_pat, d_w_pat, d_lo_rank, d_w_rank, d_cbin, d_rows, d_nck, d_w_nb, d_ibase_h, d_icoff_h, d_lo_rlo, d_lo_rhi})
    if (p) (void)hipFree(p);
}

template <class T_>
static int up(const std::vector<T_> &v, void **d) {
  DNM_HIP(hipMalloc(d, std::max<size_t>(1, v.size()) * sizeof(T_)));
  DNM_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(T_), hipMemcpyHostToDevice));
  return 0;
}

int Sc3Layout::init(int L, int k, int a, int w, bool want_device, int order_) {
  DNM_CHECK(sc3_valid(L, k, a, w), "no such vector layout: L=%d k=%d a=%d w=%d", L, k, a, w);
  DNM_CHECK(order_ == 0 || order_ == 1, "unknown block order %d of a SpinConserve layout", order_);
  order = order_;
  Sc3Tab &S = host;
  S = Sc3Tab{};
  S.L = L; S.k = k; S.a = a; S.w = w; S.t = L - a - w;
  const int t = S.t;
  cbin.assign(17 * 17, 0);
  for (int n = 0; n < 17; ++n) for (int j = 0; j < 17; ++j) cbin[n * 17 + j] = (int32_t)hbinom(n, j);
  nck.assign((size_t)(k + 1) * (L + 1), 0);
  for (int kk = 0; kk <= k; ++kk) for (int LL = 0; LL <= L; ++LL) nck[(size_t)kk * (L + 1) + LL] = hbinom(LL, kk);
  lo_rank.assign((size_t)1 << a, 0);
  w_rank.assign((size_t)1 << w, 0);
  lo_pat.clear();
  w_pat.clear();
  for (int j = 0; j <= a; ++j) {
    S.lo_off[j] = (int32_t)lo_pat.size();
    S.nl[j] = (int32_t)hbinom(a, j);
    S.pitch[j] = (S.nl[j] + 7) / 8 * 8;
    int r = 0;
    for (uint32_t v = 0; v < (1u << a); ++v)
      if (__builtin_popcount(v) == j) { lo_rank[v] = (uint16_t)r++; lo_pat.push_back((uint16_t)v); }
  }
  S.lo_off[a + 1] = (int32_t)lo_pat.size();
  const int wmax = (int)hbinom(w, w / 2);
  for (int j = 0; j <= w; ++j) {
    S.w_off[j] = (int32_t)w_pat.size();
    S.nw[j] = (int32_t)hbinom(w, j);
    int r = 0;
    for (uint32_t v = 0; v < (1u << w); ++v)
      if (__builtin_popcount(v) == j) { w_rank[v] = (uint16_t)r++; w_pat.push_back((uint16_t)v); }
    int s = 0;
    while ((S.nw[j] << (s + 1)) <= wmax && s < 8) ++s;      // R = 16 << s keeps nw * R within the largest tile
    S.rs[j] = s;
  }
  S.w_off[w + 1] = (int32_t)w_pat.size();
  // partner table of the window pass's LDS bonds (Sc3Tab::w_nb)
  // (only the field splits that have kernel instances need them; wider ones keep the tables empty)
  const bool nbfit = w - 1 <= 16 && hbinom(w, w / 2) < 255;
  w_nb.assign(nbfit ? (size_t)2 * w_pat.size() : 0, 0);
  for (int j = 0; nbfit && j <= w; ++j)
    for (int r = 0; r < S.nw[j]; ++r) {
      const uint32_t v = w_pat[S.w_off[j] + r];
      for (int b = 0; b < 16; ++b) {
        uint64_t f = (uint64_t)S.nw[j];                        // the zero row
        if (b < w - 1) {
          const uint32_t pair = (v >> b) & 3u;
          if (pair == 1u || pair == 2u) f = w_rank[v ^ (3u << b)];
        }
        w_nb[(size_t)2 * (S.w_off[j] + r) + (size_t)(b / 8)] |= f << (8 * (b % 8));
      }
    }
  // lo_rank in two halves (Sc3Tab::lo_rlo / lo_rhi): colex rank = sum over the ones, m-th one at position q: C(q, m)
  {
    const int h = a / 2, hb = a - h;
    lo_rlo.assign((size_t)1 << h, 0);
    lo_rhi.assign(((size_t)1 << hb) * (h + 1), 0);
    for (uint32_t v = 0; v < (1u << h); ++v) {
      int64_t r = 0;
      int m = 0;
      for (int q = 0; q < h; ++q) if ((v >> q) & 1u) r += hbinom(q, ++m);
      lo_rlo[v] = (uint16_t)r;
    }
    for (uint32_t v = 0; v < (1u << hb); ++v)
      for (int cl = 0; cl <= h; ++cl) {
        int64_t r = 0;
        int m = cl;
        for (int q = 0; q < hb; ++q) if ((v >> q) & 1u) r += hbinom(h + q, ++m);
        lo_rhi[(size_t)v * (h + 1) + cl] = (uint16_t)r;
      }
    for (uint32_t v = 0; v < (1u << a); ++v) {
      const uint32_t lo = v & ((1u << h) - 1u);
      DNM_CHECK(lo_rank[v] == lo_rlo[lo] + lo_rhi[(size_t)(v >> h) * (h + 1) + __builtin_popcount(lo)],
                "internal: split rank table of pattern %u", v);
    }
  }
  icoff.assign((size_t)(a + w + 1) * (w + 1), 0);
  ncoff.assign((size_t)(a + w + 1) << w, 0);
  std::vector<int64_t> isize(a + w + 1, 0);
  for (int kr = 0; kr <= a + w; ++kr) {
    int64_t o = 0;
    for (int cw = 0; cw <= w; ++cw) {
      icoff[(size_t)kr * (w + 1) + cw] = o;
      const int kl = kr - cw;
      if (kl >= 0 && kl <= a) o += hbinom(w, cw) * S.pitch[kl];
    }
    isize[kr] = o;
    int64_t no = 0;
    for (uint32_t W = 0; W < (1u << w); ++W) {
      ncoff[((size_t)kr << w) + W] = no;
      const int kl = kr - __builtin_popcount(W);
      if (kl >= 0 && kl <= a) no += hbinom(a, kl);
    }
  }
  ibase.assign((size_t)1 << t, -1);
  nbase.assign((size_t)1 << t, -1);
  rows.clear();
  int64_t ni = 0, nn = 0;
  // reference indices: ascending T
  tseq.clear();
  for (uint32_t T = 0; T < (1u << t); ++T) {
    const int kr = k - __builtin_popcount(T);
    if (kr < 0 || kr > a + w) continue;
    nbase[T] = nn;
    nn += hbinom(a + w, kr);
    tseq.push_back(T);
  }
  // the order the blocks lie in (sc3_code_order)
  if (order == 1) {
    // by (ones of T above its lowest bit, ones of T's upper half, T >> 1, T & 1): the two blocks that differ in T's lowest
    // bit -- partners under the W/T boundary bond -- lie side by side, the bonds inside T >> 1 keep the first key, and of
    // a chain's bonds only the one between T's two lowest bits changes it (sc3.h: sc3_code_order)
    const int th = t / 2;
    std::stable_sort(tseq.begin(), tseq.end(), [th](uint32_t x, uint32_t y) {
      const int px = __builtin_popcount(x >> 1), py = __builtin_popcount(y >> 1);
      if (px != py) return px < py;
      const int hx = __builtin_popcount(x >> th), hy = __builtin_popcount(y >> th);
      if (hx != hy) return hx < hy;
      return x < y;                    // (T >> 1, then T & 1)
    });
  }
  tidx.assign((size_t)1 << t, 0xffffffffu);
  rowstart.assign(tseq.size() + 1, 0);
  for (size_t b = 0; b < tseq.size(); ++b) {
    const uint32_t T = tseq[b];
    const int kr = k - __builtin_popcount(T);
    tidx[T] = (uint32_t)b;
    ibase[T] = ni;
    ni += isize[kr];
    rowstart[b] = rows.size();
    for (uint32_t W = 0; W < (1u << w); ++W) {
      const int kl = kr - __builtin_popcount(W);
      if (kl >= 0 && kl <= a) rows.push_back((T << w) | W);
    }
  }
  rowstart[tseq.size()] = rows.size();
  S.nint = ni;
  dim = nn;
  DNM_CHECK(nn == hbinom(L, k), "internal: layout does not cover the subspace");
  S.ibase = ibase.data(); S.nbase = nbase.data(); S.icoff = icoff.data(); S.ncoff = ncoff.data();
  S.lo_pat = lo_pat.data(); S.w_pat = w_pat.data(); S.lo_rank = lo_rank.data(); S.w_rank = w_rank.data();
  S.cbin = cbin.data();
  S.nck = nck.data();
  S.w_nb = w_nb.data();
  S.lo_rlo = lo_rlo.data();
  S.lo_rhi = lo_rhi.data();
  dev = S;
  // halved positions (real vectors read as pairs of entries): everything is a multiple of 8 entries
  ibase_h.assign(ibase.size(), -1);
  for (size_t i = 0; i < ibase.size(); ++i) if (ibase[i] >= 0) ibase_h[i] = ibase[i] / 2;
  icoff_h.assign(icoff.size(), 0);
  for (size_t i = 0; i < icoff.size(); ++i) icoff_h[i] = icoff[i] / 2;
  host_h = S;
  host_h.ibase = ibase_h.data();
  host_h.icoff = icoff_h.data();
  host_h.nint = S.nint / 2;
  for (int j = 0; j <= a; ++j) host_h.pitch[j] = S.pitch[j] / 2;
  if (want_device) {
    DNM_TRY(up(ibase, &d_ibase)); DNM_TRY(up(nbase, &d_nbase)); DNM_TRY(up(icoff, &d_icoff));
    DNM_TRY(up(ncoff, &d_ncoff)); DNM_TRY(up(lo_pat, &d_lo_pat)); DNM_TRY(up(w_pat, &d_w_pat));
    DNM_TRY(up(lo_rank, &d_lo_rank)); DNM_TRY(up(w_rank, &d_w_rank)); DNM_TRY(up(cbin, &d_cbin));
    DNM_TRY(up(rows, &d_rows));
    DNM_TRY(up(nck, &d_nck));
    DNM_TRY(up(w_nb, &d_w_nb));
    dev.w_nb = (const uint64_t *)d_w_nb;
    DNM_TRY(up(lo_rlo, &d_lo_rlo)); DNM_TRY(up(lo_rhi, &d_lo_rhi));
    dev.lo_rlo = (const uint16_t *)d_lo_rlo; dev.lo_rhi = (const uint16_t *)d_lo_rhi;
    dev.ibase = (const int64_t *)d_ibase; dev.nbase = (const int64_t *)d_nbase;
    dev.icoff = (const int64_t *)d_icoff; dev.ncoff = (const int64_t *)d_ncoff;
    dev.lo_pat = (const uint16_t *)d_lo_pat; dev.w_pat = (const uint16_t *)d_w_pat;
    dev.lo_rank = (const uint16_t *)d_lo_rank; dev.w_rank = (const uint16_t *)d_w_rank;
    dev.cbin = (const int32_t *)d_cbin;
    dev.nck = (const int64_t *)d_nck;
    DNM_TRY(up(ibase_h, &d_ibase_h)); DNM_TRY(up(icoff_h, &d_icoff_h));
    dev_h = dev;
    dev_h.ibase = (const int64_t *)d_ibase_h;
    dev_h.icoff = (const int64_t *)d_icoff_h;
    dev_h.nint = host_h.nint;
    for (int j = 0; j <= a; ++j) dev_h.pitch[j] = host_h.pitch[j];
    on_device = true;
  }
  return 0;
}

const Sc3Layout *sc3_get(int L, int k, int a, int w, bool want_device, int order) {
  static std::mutex mu;
  static std::map<std::array<int, 6>, std::unique_ptr<Sc3Layout>> cache;
  std::lock_guard<std::mutex> g(mu);
  const std::array<int, 6> key{L, k, a, w, want_device ? 1 : 0, order};
  auto it = cache.find(key);
  if (it != cache.end()) return it->second.get();
  std::unique_ptr<Sc3Layout> lay(new Sc3Layout());
  if (lay->init(L, k, a, w, want_device, order)) return nullptr;
  return (cache[key] = std::move(lay)).get();
}

// ---- utilities ------------------------------------------------------------------------------------------
// the rows of the T blocks [T0, T1) inside Ly.rows (sorted by T, then W) and the offsets of that range
struct RowRange { size_t first, count; int64_t ioff, noff; };
static RowRange row_range(const Sc3Layout &Ly, uint32_t T0, uint32_t T1) {
  const uint32_t nb = (uint32_t)Ly.tseq.size();
  const uint32_t b0 = std::min(T0, nb), b1 = std::max(b0, std::min(T1, nb));
  RowRange r;
  r.first = Ly.rowstart[b0];
  r.count = Ly.rowstart[b1] - Ly.rowstart[b0];
  int64_t il, nl;
  sc3_range(Ly, T0, T1, &r.ioff, &il, &r.noff, &nl);
  return r;
}
// the maps between a layout and the reference order need the range's reference side to be a range too: every range of
// block order 0, whole vectors of the others
static int ref_side(const Sc3Layout &Ly, const RowRange &r) {
  DNM_CHECK(Ly.order == 0 || (r.first == 0 && r.count == Ly.rows.size()),
            "a rank's share of a SpinConserve layout in block order %d is no range of the reference order", Ly.order);
  return 0;
}

bool sc3_perm_make(const int8_t *site_perm, int L, Sc3Perm *out) {
  *out = Sc3Perm{};
  out->L = L;
  for (int i = 0; i < 64; ++i) out->to_int[i] = out->to_ref[i] = (uint8_t)i;
  if (!site_perm) return true;
  uint64_t seen = 0;
  for (int i = 0; i < L; ++i) {
    const int b = site_perm[i];
    if (b < 0 || b >= L || ((seen >> b) & 1ull)) return false;
    seen |= 1ull << b;
    out->to_int[i] = (uint8_t)b;
    out->to_ref[b] = (uint8_t)i;
    if (b != i) out->on = 1;
  }
  return true;
}

// a relabelled layout covers whole vectors on one rank
static int perm_whole(const Sc3Layout &Ly, const Sc3Perm *perm, const RowRange &r) {
  // (whole vectors, or the blocks below a bound -- the half whose top bit is clear, an XParity vector: both start at
  // position 0 of the layout and at index 0 of the reference order)
  DNM_CHECK(!perm || !perm->on || (r.first == 0 && r.ioff == 0 && r.noff == 0),
            "a relabelled SpinConserve layout is not partitioned over ranks");
  return 0;
}

int sc3_layout_copy(const Sc3Layout &Ly, void *dst, const void *src, bool to_internal, hipStream_t st, uint32_t T0, uint32_t T1,
                    const Sc3Perm *perm) {
  DNM_CHECK(Ly.on_device, "layout tables are not on the device");
  const RowRange r = row_range(Ly, T0, T1);
  if (!r.count) return 0;
  DNM_TRY(perm_whole(Ly, perm, r));
  DNM_TRY(ref_side(Ly, r));
  if (perm && perm->on) {
    hipLaunchKernelGGL(sc3_copy_perm_kernel<c128>, dim3((unsigned)r.count), dim3(256), 0, st, Ly.dev, *perm,
                       (const uint32_t *)Ly.d_rows, (c128 *)dst, (const c128 *)src, to_internal ? 1 : 0);
    DNM_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(sc3_copy_kernel<c128>, dim3((unsigned)r.count), dim3(256), 0, st, Ly.dev,
                     (const uint32_t *)Ly.d_rows + r.first, (c128 *)dst, (const c128 *)src, to_internal ? 1 : 0, r.ioff, r.noff);
  DNM_HIP(hipGetLastError());
  return 0;
}
int sc3_layout_copy_f64(const Sc3Layout &Ly, double *dst, const double *src, bool to_internal, hipStream_t st, uint32_t T0,
                        uint32_t T1, const Sc3Perm *perm) {
  DNM_CHECK(Ly.on_device, "layout tables are not on the device");
  const RowRange r = row_range(Ly, T0, T1);
  if (!r.count) return 0;
  DNM_TRY(perm_whole(Ly, perm, r));
  DNM_TRY(ref_side(Ly, r));
  if (perm && perm->on) {
    hipLaunchKernelGGL(sc3_copy_perm_kernel<double>, dim3((unsigned)r.count), dim3(256), 0, st, Ly.dev, *perm,
                       (const uint32_t *)Ly.d_rows, dst, src, to_internal ? 1 : 0);
    DNM_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(sc3_copy_kernel<double>, dim3((unsigned)r.count), dim3(256), 0, st, Ly.dev,
                     (const uint32_t *)Ly.d_rows + r.first, dst, src, to_internal ? 1 : 0, r.ioff, r.noff);
  DNM_HIP(hipGetLastError());
  return 0;
}
int sc3_zero_padding(const Sc3Layout &Ly, void *x, hipStream_t st, uint32_t T0, uint32_t T1) {
  DNM_CHECK(Ly.on_device, "layout tables are not on the device");
  const RowRange r = row_range(Ly, T0, T1);
  if (!r.count) return 0;
  hipLaunchKernelGGL(sc3_zero_pad_kernel, dim3((unsigned)r.count), dim3(64), 0, st, Ly.dev,
                     (const uint32_t *)Ly.d_rows + r.first, (c128 *)x, r.ioff);
  DNM_HIP(hipGetLastError());
  return 0;
}
int sc3_random(const Sc3Layout &Ly, void *x, uint64_t seed, hipStream_t st, uint32_t T0, uint32_t T1, const Sc3Perm *perm) {
  DNM_CHECK(Ly.on_device, "layout tables are not on the device");
  const RowRange r = row_range(Ly, T0, T1);
  if (!r.count) return 0;
  DNM_TRY(perm_whole(Ly, perm, r));
  if (perm && perm->on) {
    hipLaunchKernelGGL(sc3_random_perm_kernel<false>, dim3((unsigned)r.count), dim3(256), 0, st, Ly.dev, *perm,
                       (const uint32_t *)Ly.d_rows, x, seed);
    DNM_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(sc3_random_kernel, dim3((unsigned)r.count), dim3(256), 0, st, Ly.dev,
                     (const uint32_t *)Ly.d_rows + r.first, (c128 *)x, seed, r.ioff);
  DNM_HIP(hipGetLastError());
  return 0;
}
int sc3_random_real(const Sc3Layout &Ly, double *x, uint64_t seed, hipStream_t st, uint32_t T0, uint32_t T1,
                    const Sc3Perm *perm) {
  DNM_CHECK(Ly.on_device, "layout tables are not on the device");
  const RowRange r = row_range(Ly, T0, T1);
  if (!r.count) return 0;
  DNM_TRY(perm_whole(Ly, perm, r));
  if (perm && perm->on) {
    hipLaunchKernelGGL(sc3_random_perm_kernel<true>, dim3((unsigned)r.count), dim3(256), 0, st, Ly.dev, *perm,
                       (const uint32_t *)Ly.d_rows, (void *)x, seed);
    DNM_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(sc3_random_real_kernel, dim3((unsigned)r.count), dim3(256), 0, st, Ly.dev,
                     (const uint32_t *)Ly.d_rows + r.first, x, seed, r.ioff);
  DNM_HIP(hipGetLastError());
  return 0;
}
int sc3_unpack_real(const Sc3Layout &Ly, void *dst, const double *src, hipStream_t st, uint32_t T0, uint32_t T1) {
  int64_t is, n, ns, nl;
  sc3_range(Ly, T0, T1, &is, &n, &ns, &nl);
  if (n <= 0) return 0;
  const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)1 << 22);
  hipLaunchKernelGGL(sc3_unpack_real_kernel, dim3(nb), dim3(256), 0, st, (c128 *)dst, src, n);
  DNM_HIP(hipGetLastError());
  return 0;
}
int sc3_positions(const Sc3Layout &Ly, int64_t n, const int64_t *idx, int64_t *pos, hipStream_t st, uint32_t T0, uint32_t T1,
                  const Sc3Perm *perm) {
  DNM_CHECK(Ly.on_device, "layout tables are not on the device");
  if (n <= 0) return 0;
  DNM_CHECK(n < ((int64_t)1 << 32), "more than 2^32 indices in one call (one thread each: a launch holds fewer)");
  const RowRange r = row_range(Ly, T0, T1);
  DNM_TRY(perm_whole(Ly, perm, r));
  DNM_TRY(ref_side(Ly, r));
  if (perm && perm->on) {
    hipLaunchKernelGGL(sc3_positions_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Ly.dev, *perm, n, idx, pos);
    DNM_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(sc3_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Ly.dev, n, idx, pos, r.ioff,
                     r.noff);
  DNM_HIP(hipGetLastError());
  return 0;
}

// ---- the operator ---------------------------------------------------------------------------------------
// deal groups of workgroups to the 8 XCDs (workgroup b runs on XCD b % 8): the next group goes to the shortest stream
static std::vector<uint32_t> deal(const std::vector<std::vector<uint32_t>> &groups) {
  std::vector<std::vector<uint32_t>> st(8);
  for (auto &g : groups) {
    int best = 0;
    for (int s = 1; s < 8; ++s) if (st[s].size() < st[best].size()) best = s;
    st[best].insert(st[best].end(), g.begin(), g.end());
  }
  size_t n = 0;
  for (auto &s : st) n = std::max(n, s.size());
  std::vector<uint32_t> out(8 * n, 0xffffffffu);
  for (int s = 0; s < 8; ++s)
    for (size_t i = 0; i < st[s].size(); ++i) out[8 * i + s] = st[s][i];
  return out;
}

// lo pass: rows -> workgroups.  `order` is the dispatch order of the rows (8 interleaved XCD streams, 0xffffffff =
// padding); inside a stream rows are packed, in that order, 2^m to a workgroup where 2^m rows of their length fit the
// NT * RPT entries a workgroup's threads hold (sub-groups of whole wavefronts, m <= 3).  Rows that follow each other in
// a stream are the same (cw, wr) over the T's of a popcount class, so a workgroup's rows have equal lengths and its
// place in the stream stays next to the boundary-bond partners of its rows.  Returns 8 entries per workgroup, the
// workgroups of the streams interleaved again.
static std::vector<uint32_t> pack_lo_rows(const std::vector<uint32_t> &order, const Sc3Layout &ly, int nt, bool real) {
  const Sc3Tab &S = ly.host;
  const int cap = real ? sc3_lo_cap_r(S.a, nt) : sc3_lo_cap(S.a, nt);      // entries a workgroup's threads hold
  if (real) nt = sc3r_threads(nt);
  int maxm = 0;
  while (maxm < 3 && (nt >> (maxm + 1)) >= 64) ++maxm;
  auto logm_of = [&](uint32_t e) {
    const uint32_t T = e >> S.w, W = e & ((1u << S.w) - 1u);
    const int kl = S.k - __builtin_popcount(T) - __builtin_popcount(W);
    int m = 0;
    while (m < maxm && S.nl[kl] <= (cap >> (m + 1))) ++m;
    return m;
  };
  std::vector<std::vector<uint32_t>> wgs(8);        // per stream: 8 entries per workgroup
  for (int s = 0; s < 8; ++s) {
    std::vector<uint32_t> open[4];                  // rows waiting for their workgroup to fill, by m
    auto flush = [&](int m) {
      if (open[m].empty()) return;
      for (int j = 0; j < 8; ++j)
        wgs[s].push_back(j < (int)open[m].size() ? (open[m][j] | ((uint32_t)m << 30)) : (SC3_NOROW | ((uint32_t)m << 30)));
      open[m].clear();
    };
    for (size_t i = s; i < order.size(); i += 8) {
      const uint32_t e = order[i];
      if (e == 0xffffffffu) continue;
      const int m = logm_of(e);
      open[m].push_back(e);
      if ((int)open[m].size() == (1 << m)) flush(m);
    }
    for (int m = 0; m < 4; ++m) flush(m);
  }
  size_t n = 0;
  for (auto &v : wgs) n = std::max(n, v.size() / 8);
  std::vector<uint32_t> out(8 * 8 * n, 0xffffffffu);
  for (int s = 0; s < 8; ++s)
    for (size_t i = 0; i < wgs[s].size() / 8; ++i)
      for (int j = 0; j < 8; ++j) out[8 * (8 * i + s) + j] = wgs[s][8 * i + j];
  return out;
}

bool sc3_instance(int a, int w) { return (a == 14 && w == 10) || (a == 6 && w == 4); }

Sc3Mat::~Sc3Mat() {
  for (void *p : {d_permA, d_permB, d_bond, d_dlo, d_dt_sign, d_dt_coef, d_dt_group, d_rowsel, d_hops, d_wnb, d_ptab, d_pcoef})
    if (p) (void)hipFree(p);
}

static int64_t block_len(const Sc3Layout &ly, uint32_t T) {        // internal length of the T block
  if (ly.ibase[T] < 0) return 0;
  const uint32_t b = ly.tidx[T];                                     // (the next block of the layout's sequence, whatever its order)
  return (b + 1 < (uint32_t)ly.tseq.size() ? ly.ibase[ly.tseq[b + 1]] : ly.host.nint) - ly.ibase[T];
}

std::vector<uint32_t> sc3_partition(const Sc3Layout &ly, int nranks) {
  const uint32_t nb = (uint32_t)ly.tseq.size();
  std::vector<uint32_t> Tb((size_t)nranks + 1, nb);
  Tb[0] = 0;
  uint32_t b = 0;
  for (int r = 1; r < nranks; ++r) {
    const int64_t target = (int64_t)((__int128)ly.host.nint * r / nranks);
    while (b < nb && ly.ibase[ly.tseq[b]] < target) ++b;
    Tb[r] = b;
  }
  return Tb;
}

void sc3_range(const Sc3Layout &ly, uint32_t T0, uint32_t T1, int64_t *istart, int64_t *ilen, int64_t *nstart, int64_t *nlen) {
  const uint32_t nb = (uint32_t)ly.tseq.size();
  const uint32_t b0 = std::min(T0, nb), b1 = std::max(b0, std::min(T1, nb));
  const int64_t i0 = b0 < nb ? ly.ibase[ly.tseq[b0]] : ly.host.nint, i1 = b1 < nb ? ly.ibase[ly.tseq[b1]] : ly.host.nint;
  *istart = i0;
  *ilen = i1 - i0;
  if (ly.order == 0) {
    const int64_t n0 = b0 < nb ? ly.nbase[ly.tseq[b0]] : ly.dim, n1 = b1 < nb ? ly.nbase[ly.tseq[b1]] : ly.dim;
    *nstart = n0;
    *nlen = n1 - n0;
    return;
  }
  // any other block order: the states of the range (its reference side is a range only for the whole sequence)
  int64_t n = 0;
  const int aw = ly.host.a + ly.host.w;
  for (uint32_t b = b0; b < b1; ++b) n += hbinom(aw, ly.host.k - __builtin_popcount(ly.tseq[b]));
  *nstart = b0 == 0 ? 0 : -1;
  *nlen = n;
}

void Sc3Mat::window(int64_t *lo, int64_t *hi) const {
  int64_t a = INT64_MAX, b = INT64_MIN;
  for (uint32_t T = 0; T < (uint32_t)needT.size(); ++T)
    if (needT[T] && ly->ibase[T] >= 0) {
      a = std::min(a, ly->ibase[T]);
      b = std::max(b, ly->ibase[T] + block_len(*ly, T) - 1);
    }
  if (b < a) a = b = row0;
  *lo = a;
  *hi = b;
}

// the needed blocks as maximal runs of positions [lo, hi), ascending
std::vector<std::pair<int64_t, int64_t>> Sc3Mat::ranges() const {
  std::vector<std::pair<int64_t, int64_t>> blk;
  for (uint32_t T = 0; T < (uint32_t)needT.size(); ++T)
    if (needT[T] && ly->ibase[T] >= 0) blk.push_back({ly->ibase[T], ly->ibase[T] + block_len(*ly, T)});
  std::sort(blk.begin(), blk.end());
  std::vector<std::pair<int64_t, int64_t>> out;
  for (const auto &b : blk) {
    if (!out.empty() && out.back().second == b.first) out.back().second = b.second;
    else out.push_back(b);
  }
  return out;
}

void Sc3Mat::chunks(int shift, int64_t first_chunk, int64_t nchunks, uint8_t *map) const {
  for (int64_t c = 0; c < nchunks; ++c) map[c] = 0;
  for (uint32_t T = 0; T < (uint32_t)needT.size(); ++T)
    if (needT[T] && ly->ibase[T] >= 0) {
      const int64_t c0 = (ly->ibase[T] >> shift) - first_chunk, c1 = ((ly->ibase[T] + block_len(*ly, T) - 1) >> shift) - first_chunk;
      for (int64_t c = std::max<int64_t>(c0, 0); c <= c1 && c < nchunks; ++c) map[c] = 1;
    }
}

int Sc3Mat::init(const Sc3Layout *layout, const std::vector<int64_t> &masks, const std::vector<int64_t> &mask_offsets,
                 const std::vector<int64_t> &signs, const std::vector<double> &rcoef, const std::vector<ScMask> &scm,
                 bool want_device, uint32_t T0_, uint32_t T1_, bool real_vectors) {
  ly = layout;
  real = real_vectors;
  T0 = T0_;
  T1 = T1_;
  const Sc3Tab &S = ly->host;
  const int L = S.L, a = S.a, w = S.w, t = S.t, k = S.k;
  const int64_t nmasks = (int64_t)masks.size();
  {
    int64_t il, ns, nl;
    sc3_range(*ly, T0, T1, &row0, &il, &ns, &nl);
  }
  rowsel.clear();
  {
    const uint32_t nb = (uint32_t)ly->tseq.size();
    const uint32_t b0 = std::min(T0, nb), b1 = std::max(b0, std::min(T1, nb));
    rowsel.assign(ly->rows.begin() + (ptrdiff_t)ly->rowstart[b0], ly->rows.begin() + (ptrdiff_t)ly->rowstart[b1]);
  }
  if (rowsel.empty()) rowsel.push_back(0xffffffffu);
  if (want_device) DNM_TRY(up(rowsel, &d_rowsel));
  // the T blocks these rows read: their own and, for every mask that flips bits of T, the partner's
  needT.assign((size_t)1 << t, 0);
  for (uint32_t bq = T0; bq < T1 && bq < (uint32_t)ly->tseq.size(); ++bq) {
    const uint32_t T = ly->tseq[bq];
    needT[T] = 1;
    for (int64_t m = 0; m < nmasks; ++m) {
      const uint64_t hm = (uint64_t)masks[m] >> (a + w);
      const uint32_t U = T ^ (uint32_t)hm;
      if (!hm || U >= (1u << t) || ly->ibase[U] < 0) continue;     // (a mask that leaves T alone reads T itself)
      // a mask that flips bits of T only keeps the state in the subspace only if it keeps T's popcount
      const bool inside_T = ((uint64_t)masks[m] & (((uint64_t)1 << (a + w)) - 1)) == 0;
      if (inside_T && __builtin_popcount(U) != __builtin_popcount(T)) continue;
      needT[U] = 1;
    }
  }
  // Two tiled passes need every off-diagonal mask to be a pair hop with signs inside the pair (ScMask::pair); masks
  // that never keep a state in the subspace (an odd number of flips: the fields of the harness's long-range model)
  // are skipped.  Chains of adjacent spins take the kernels of this file, any other bond graph those of
  // sc3g_kernels.hip (DNM_SC3_GRAPH=1: chains as well, for A/B runs).
  bool chain = sc3_instance(a, w), pairs = sc3_instance(a, w);
  for (int64_t m = 0; m < nmasks; ++m) {
    if (masks[m] == 0 || scm[m].dead) continue;
    if (!scm[m].fast) chain = false;
    if (!scm[m].pair) pairs = false;
  }
  if (const char *e = knob("DNM_SC3_GRAPH")) if (e[0] == '1') chain = false;
  tiled = chain || pairs;
  graph = tiled && !chain;
  if (!tiled) return 0;
  std::vector<double> bond(4 * (size_t)std::max(1, L - 1), 0.0);
  op.present = 0;
  sym = true;
  for (int64_t m = 0; m < nmasks; ++m) {
    if (masks[m] == 0 || scm[m].dead) continue;
    if (scm[m].up_im != 0.0 || scm[m].dn_im != 0.0 || scm[m].up_re != scm[m].dn_re) sym = false;
    if (graph) continue;
    const int b = scm[m].lo;
    bond[4 * b] = scm[m].up_re; bond[4 * b + 1] = scm[m].up_im;
    bond[4 * b + 2] = scm[m].dn_re; bond[4 * b + 3] = scm[m].dn_im;
    op.present |= 1ull << b;
  }
  // which pass gathers which bond outside its LDS tile: the Lo/W boundary in the lo pass, the W/T boundary and the
  // bonds inside T in the window pass (measured, profiles/r03_exp3_sc3_v2.txt)
  op.bondsA = op.present & (1ull << (a - 1));
  op.bondsB = 0;
  for (int b = a + w - 1; b < L - 1; ++b) op.bondsB |= op.present & (1ull << b);
  // any bond graph: the hops by pass and by the way they are applied
  hops.clear();
  wnb.clear();
  size_t nh[4] = {0, 0, 0, 0};
  if (graph) {
    std::vector<Sc3Hop> part[4];       // lds A, gathered A, lds B, gathered B
    auto field = [&](int b) { return b < a ? 0 : (b < a + w ? 1 : 2); };
    auto fstart = [&](int f) { return f == 0 ? 0 : (f == 1 ? a : a + w); };
    for (int64_t m = 0; m < nmasks; ++m) {
      if (masks[m] == 0 || scm[m].dead) continue;
      const uint64_t mk = (uint64_t)masks[m];
      Sc3Hop h{};
      h.mLo = (uint32_t)(mk & (((uint64_t)1 << a) - 1));
      h.mW = (uint32_t)((mk >> a) & (((uint64_t)1 << w) - 1));
      h.mT = (uint32_t)(mk >> (a + w));
      h.half = __builtin_popcountll(mk) / 2;
      const int fi = field(scm[m].lo), fj = field(scm[m].hi);
      h.dfield = scm[m].pair == 2 ? 3 : fi;      // (the flip-composed hops of XParity act one way only)
      h.dbit = scm[m].lo - fstart(fi);
      h.up_re = scm[m].up_re; h.up_im = scm[m].up_im; h.dn_re = scm[m].dn_re; h.dn_im = scm[m].dn_im;
      part[fi == 0 ? (fj == 0 ? 0 : 1) : (fi == 1 && fj == 1 ? 2 : 3)].push_back(h);
    }
    // probes (timing only, WRONG results): keep the first n hops of a kind -- what a pass would take without the others
    // bounds what any reworking of them can gain (tools/probes/sc3g_drop_hops.sh)
    for (int q = 0; q < 4; ++q) {
      static const char *names[4] = {"DNM_SC3G_KEEP_LDSA", "DNM_SC3G_KEEP_GATA", "DNM_SC3G_KEEP_LDSB", "DNM_SC3G_KEEP_GATB"};
      if (const char *e = knob(names[q]))
        if ((size_t)atoi(e) < part[q].size()) part[q].resize((size_t)atoi(e));
    }
    if ((int)part[1].size() > SC3G_MAX_GATHER || (int)part[3].size() > SC3G_MAX_GATHER ||
        (int)part[2].size() > SC3G_MAX_WLDS) {      // more hops than a pass has lanes / table columns for: the row kernel
      tiled = graph = false;
      return 0;
    }
    for (int q = 0; q < 4; ++q) {
      nh[q] = part[q].size();
      hops.insert(hops.end(), part[q].begin(), part[q].end());
    }
    if (hops.empty()) hops.push_back(Sc3Hop{});
    // partner rows of the window pass's LDS hops
    const size_t nb = nh[2];
    wnb.assign(std::max<size_t>(1, ly->w_pat.size() * nb), 0);
    for (int cw = 0; cw <= w; ++cw)
      for (int wr = 0; wr < S.nw[cw]; ++wr) {
        const uint32_t v = ly->w_pat[S.w_off[cw] + wr];
        for (size_t q = 0; q < nb; ++q) {
          const uint32_t mw = part[2][q].mW;
          wnb[(size_t)(S.w_off[cw] + wr) * nb + q] =
              (uint8_t)(__builtin_popcount(v & mw) == part[2][q].half ? ly->w_rank[v ^ mw] : S.nw[cw]);
        }
      }
  }
  op.nldsA = (int32_t)nh[0]; op.ngatA = (int32_t)nh[1]; op.nldsB = (int32_t)nh[2]; op.ngatB = (int32_t)nh[3];
  // partner table of the lo pass's LDS hops (Sc3Op::ptab).  Needs the zero entry behind a row's entries inside the
  // row's share of the tile: true for every class of the instances (C(a, kl) is no power of two above 1).
  ptab.clear();
  op.ptab = nullptr;
  op.nhp = 0;
  {
    const char *pe = knob("DNM_SC3G_PTAB");
    const int ntl = a == 14 ? 1024 : 256;          // threads of the lo pass (launch_sc3g)
    const int cap = real ? sc3_lo_cap_r(a, ntl) : sc3_lo_cap(a, ntl);
    const int nthr = real ? sc3r_threads(ntl) : ntl;
    int maxm = 0;
    while (maxm < 3 && (nthr >> (maxm + 1)) >= 64) ++maxm;
    bool ok = graph && nh[0] > 0 && (int)nh[0] <= SC3G_MAX_PTAB && !(pe && pe[0] == '0');
    for (int kl = 0; kl <= a && ok; ++kl) {
      int m = 0;
      while (m < maxm && S.nl[kl] <= (cap >> (m + 1))) ++m;
      if (S.nl[kl] >= (cap >> m) || (S.nl[kl] + 1) * (real ? 8 : 16) > 0xffff) ok = false;
    }
    if (ok) {
      const int nhp = ((int)nh[0] + 7) & ~7, esz = real ? 8 : 16;
      int row = 0;
      for (int kl = 0; kl <= a; ++kl) {
        op.ptab_row[kl] = row;
        row += (S.nl[kl] + 2) & ~1;
      }
      ptab.assign((size_t)row * nhp, 0);
      for (int kl = 0; kl <= a; ++kl) {
        const uint16_t zero = (uint16_t)(S.nl[kl] * esz);
        const int nrow = (S.nl[kl] + 2) & ~1;
        for (int r = 0; r < nrow; ++r) {
          uint16_t *t = ptab.data() + (size_t)(op.ptab_row[kl] + r) * nhp;
          for (int q = 0; q < nhp; ++q) t[q] = zero;
          if (r >= S.nl[kl]) continue;
          const uint32_t v = ly->lo_pat[S.lo_off[kl] + r];
          for (size_t q = 0; q < nh[0]; ++q) {
            const Sc3Hop &h = hops[q];               // (the LDS hops of the lo pass come first)
            if (__builtin_popcount(v & h.mLo) == h.half) t[q] = (uint16_t)(ly->lo_rank[v ^ h.mLo] * esz);
          }
        }
      }
      op.nhp = nhp;
      pcoef.assign((size_t)nhp, 0.0);
      for (size_t q = 0; q < nh[0]; ++q) pcoef[q] = hops[q].up_re;
    }
  }
  // diagonal on the fly: split the mask-0 terms by what their sign masks see
  diag_mode = 0;
  std::vector<double> dlo;
  std::vector<uint64_t> dt_sign;
  std::vector<double> dt_coef;
  std::vector<int32_t> dt_group;
  op.ngroups = 0;
  if (nmasks > 0 && masks[0] == 0) {
    diag_mode = 2;
    const uint64_t lom = ((uint64_t)1 << a) - 1;
    dlo.assign(ly->lo_pat.size(), 0.0);
    std::vector<uint64_t> groups;
    for (int64_t tt = mask_offsets[0]; tt < mask_offsets[1]; ++tt) {
      const uint64_t sg = (uint64_t)signs[tt];
      const double c = rcoef[tt];
      if ((sg & ~lom) == 0) {
        for (size_t i = 0; i < ly->lo_pat.size(); ++i)
          dlo[i] += (__builtin_popcountll(ly->lo_pat[i] & sg) & 1) ? -c : c;
        continue;
      }
      int g = 0;
      if (sg & lom) {
        size_t j = 0;
        while (j < groups.size() && groups[j] != (sg & lom)) ++j;
        if (j == groups.size()) groups.push_back(sg & lom);
        g = (int)j + 1;
      }
      dt_sign.push_back((sg >> a) | ((uint64_t)g << 61));
      dt_coef.push_back(c);
      dt_group.push_back(g);
    }
    if (groups.size() > 4) diag_mode = 1;      // too many mixed patterns: the cached diagonal instead
    else {
      op.ngroups = (int32_t)groups.size();
      for (size_t j = 0; j < groups.size(); ++j) op.glo[j] = (uint32_t)groups[j];
    }
  }
  op.ndt = diag_mode == 2 ? (int32_t)dt_sign.size() : 0;
  // dispatch order: workgroups that gather from each other run on one XCD at one time (their requests meet in that
  // XCD's L2).  Window pass: groups (kt, cw, run) over the T's of a popcount class -- siblings under the T bonds.
  std::vector<std::vector<uint32_t>> Tby(t + 1), gA, gB;
  for (uint32_t bq = T0; bq < T1 && bq < (uint32_t)ly->tseq.size(); ++bq) Tby[__builtin_popcount(ly->tseq[bq])].push_back(ly->tseq[bq]);
  for (auto &v : Tby) std::sort(v.begin(), v.end());
  for (int kt = 0; kt <= t; ++kt) {
    if (Tby[kt].empty()) continue;
    const int kr = k - kt;
    for (int cw = 0; cw <= w; ++cw) {
      const int kl = kr - cw;
      if (kl < 0 || kl > a) continue;
      // lo pass: it gathers the Lo/W boundary bond only, which couples the rows (T, W) and (T, W ^ 1): each pair goes to
      // one XCD back to back, so that what one row gathers is what the other stages (their requests meet in the L2);
      // pairs of one (cw, wr) over the T's of the class follow each other -- equal lengths side by side
      for (int wr = 0; wr < S.nw[cw]; ++wr) {
        const uint32_t W = ly->w_pat[S.w_off[cw] + wr];
        const int klp = (W & 1u) ? kl + 1 : kl - 1;                 // Lo ones of the partner row (T, W ^ 1)
        const bool partner = klp >= 0 && klp <= a;
        if ((W & 1u) && partner) continue;                          // listed with its even partner
        for (uint32_t T : Tby[kt]) {
          std::vector<uint32_t> g{(T << w) | W};
          if (partner) g.push_back((T << w) | (W ^ 1u));
          gA.push_back(g);
        }
      }
      // (real vectors: the window pass runs on pairs of entries, rows of pitch / 2 elements)
      const int Rr = 16 << S.rs[cw], nrun = ((real ? S.pitch[kl] / 2 : S.pitch[kl]) + Rr - 1) / Rr;
      DNM_CHECK(nrun < 4096, "internal: too many runs");
      for (int run = 0; run < nrun; ++run) {
        std::vector<uint32_t> g;
        for (uint32_t T : Tby[kt]) g.push_back((T << 16) | (cw << 12) | run);
        gB.push_back(g);
      }
    }
  }
  // Bond graphs: the lo pass gathers, for every hop between Lo and W, from the row (T, W ^ bit) -- rows that differ
  // in the window bits such hops touch go to one XCD back to back (what one of them gathers is what another stages:
  // the requests meet in that XCD's L2), and the groups of one window pattern over all T's follow each other, so that
  // the partner blocks of the hops between Lo and T are at least in the Infinity Cache.  DNM_SC3G_ORDER=0: the chain's
  // order (pairs under window bit 0).
  if (graph && !(knob("DNM_SC3G_ORDER") && knob("DNM_SC3G_ORDER")[0] == '0')) {
    // the bits of a row's id (T << w | W) its gathered hops flip, by the number of hops that flip them; the six most
    // used ones span a group (DNM_SC3G_ORDER=w: window bits only, the first form of this order)
    const bool wonly = knob("DNM_SC3G_ORDER") && knob("DNM_SC3G_ORDER")[0] == 'w';
    std::vector<std::pair<int, int>> use;                        // (-count, bit)
    for (int b = 0; b < t + w; ++b) {
      int cnt = 0;
      for (size_t q = nh[0]; q < nh[0] + nh[1]; ++q) {
        const uint64_t fl = ((uint64_t)hops[q].mT << w) | hops[q].mW;
        if (__builtin_popcountll(fl) == 1 && ((fl >> b) & 1ull) && !(wonly && b >= w)) ++cnt;
      }
      if (cnt) use.push_back({-cnt, b});
    }
    std::sort(use.begin(), use.end());
    uint32_t jm = 0;
    for (size_t q = 0; q < use.size() && q < 6; ++q) jm |= 1u << use[q].second;     // at most 64 rows to a group (what an XCD holds)
    gA.clear();
    std::vector<uint32_t> subs;
    for (uint32_t sset = jm;; sset = (sset - 1) & jm) {           // the subsets of jm, descending
      subs.push_back(sset);
      if (!sset) break;
    }
    std::reverse(subs.begin(), subs.end());
    const uint32_t wm = (1u << w) - 1u;
    for (uint32_t W0 = 0; W0 < (1u << w); ++W0) {
      if (W0 & jm & wm) continue;
      for (uint32_t Tb = 0; Tb < (1u << t); ++Tb) {
        if ((Tb << w) & jm) continue;
        std::vector<uint32_t> g;
        for (uint32_t sset : subs) {
          const uint32_t id = ((Tb << w) | W0) | sset, T = id >> w, W = id & wm;
          if (!ly->in_range(T, T0, T1)) continue;
          const int kl = k - __builtin_popcount(T) - __builtin_popcount(W);
          if (kl >= 0 && kl <= a) g.push_back(id);
        }
        if (!g.empty()) gA.push_back(g);
      }
    }
  }
  permA = pack_lo_rows(deal(gA), *ly, a == 14 ? 1024 : 256, real);       // thread counts of launch_sc3's instances
  // Window pass: a hop between W and T (the chain's W/T boundary bond; any such pair of a bond graph) couples the class
  // (T, cw) to (T ^ bit, cw -+ 1) at the same columns -- the same number of ones in Lo.  Workgroups of one Lo population
  // and one block of 64 columns form a group, ordered by their first column inside it, so that such partners run on one
  // XCD at about the same time (the first order grouped the T's of one popcount class at fixed (cw, run): partners under
  // the hops inside T only, which this order keeps together as well).  kagome-30: the pass's fetch 59.6 -> 28.6 B/row, L2
  // hits 49 -> 70 %, 2.40 -> 2.22 ms (profiles/r05_kagome_window_order.txt); chains: SpinConserve(32,16) 35.8 -> 30.5
  // B/row, 5.30 -> 5.07 ms, a rank of config 5 24.6 -> 24.2 ms (profiles/r05_chain_window_order.txt).
  // DNM_SC3G_WORDER=0: the first order.
  const char *worder = knob("DNM_SC3G_WORDER");
  if (!(worder && worder[0] == '0')) {
    struct Wg { uint32_t e; int kl, col; };
    std::vector<Wg> all;
    for (auto &g : gB)
      for (uint32_t e : g) {
        const uint32_t T = e >> 16;
        const int cw = (e >> 12) & 15, run = e & 0xfff;
        all.push_back({e, k - __builtin_popcount(T) - cw, run * (16 << S.rs[cw])});
      }
    int bs = 6;                                        // log2 of the column block (2^5 ... 2^7 level, 2^8: +1.5 %, 2^10: +6 %)
    if (const char *e = knob("DNM_SC3G_WBLOCK")) bs = atoi(e);
    std::stable_sort(all.begin(), all.end(), [bs](const Wg &x, const Wg &y) {
      if (x.kl != y.kl) return x.kl < y.kl;
      if ((x.col >> bs) != (y.col >> bs)) return (x.col >> bs) < (y.col >> bs);
      return x.col < y.col;
    });
    gB.clear();
    for (size_t i = 0; i < all.size();) {
      size_t j = i;
      std::vector<uint32_t> g;
      while (j < all.size() && all[j].kl == all[i].kl && (all[j].col >> bs) == (all[i].col >> bs)) g.push_back(all[j++].e);
      gB.push_back(g);
      i = j;
    }
  }
  permB = deal(gB);
  if (permA.empty()) permA.assign(8, 0xffffffffu);
  if (permB.empty()) permB.assign(8, 0xffffffffu);
  if (want_device) {
    DNM_TRY(up(permA, &d_permA)); DNM_TRY(up(permB, &d_permB)); DNM_TRY(up(bond, &d_bond));
    op.bond = (const double *)d_bond;
    if (graph) {
      DNM_TRY(up(hops, &d_hops)); DNM_TRY(up(wnb, &d_wnb));
      const Sc3Hop *hp = (const Sc3Hop *)d_hops;
      op.ldsA = hp; op.gatA = hp + nh[0]; op.ldsB = hp + nh[0] + nh[1]; op.gatB = hp + nh[0] + nh[1] + nh[2];
      op.wnb = (const uint8_t *)d_wnb;
      if (!ptab.empty()) {
        DNM_TRY(up(ptab, &d_ptab));
        op.ptab = (const uint16_t *)d_ptab;
        DNM_TRY(up(pcoef, &d_pcoef));
        op.pcoef = (const double *)d_pcoef;
      }
    }
    if (diag_mode == 2) {
      DNM_TRY(up(dlo, &d_dlo)); DNM_TRY(up(dt_sign, &d_dt_sign)); DNM_TRY(up(dt_coef, &d_dt_coef));
      DNM_TRY(up(dt_group, &d_dt_group));
      op.dlo = (const double *)d_dlo; op.dt_sign = (const uint64_t *)d_dt_sign;
      op.dt_coef = (const double *)d_dt_coef; op.dt_group = (const int32_t *)d_dt_group;
    }
  }
  return 0;
}

// phase 0: the whole multiply (window pass writes y, lo pass adds: one rank); phase 1: the part that needs nothing
// from other ranks (lo pass, writes y); phase 2: the rest (window pass, adds)
template <int A, int W, int NT, int NTW>      // NT: threads of the lo pass (512 x 7 entries: 7.4 ms against 6.4), NTW: of the window pass
static int launch_two_pass(const Sc3Mat &M, const Sc3Call &call, const double *cached_diag, const void *xw, void *y,
                           hipStream_t st, int phase) {
  const Sc3Tab &S = M.ly->dev;
  // LDS: the lo pass's row; the window pass's tile plus its zero row (largest over the classes)
  constexpr size_t ldsA = (size_t)sc3_lo_cap(A, NT) * 16;
  size_t ldsB = 0;
  for (int cw = 0; cw <= W; ++cw)      // ... and the class's partner table
    ldsB = std::max(ldsB, (((size_t)M.ly->host.nw[cw] + 1) << (4 + M.ly->host.rs[cw] + 4)) + (size_t)M.ly->host.nw[cw] * 16);
  Sc3Op op = M.op;
  const int dm = M.diag_mode;       // 2: on the fly whether or not a cached copy exists (8 B/row less to read)
  if (dm == 1) op.diag = cached_diag;
  DNM_CHECK(dm != 1 || op.diag, "this operator needs its diagonal precomputed (dnm_mat_precompute_diagonal)");
  using kern_t = void (*)(const Sc3Tab, const Sc3Op, const uint32_t *, const Sc3Call, const c128 *, c128 *);
  const bool lo_first = phase != 0;
  kern_t kB = nullptr, kA = nullptr;
  if (lo_first) kB = M.sym ? sc3_win_pass<W, NTW, true, true> : sc3_win_pass<W, NTW, false, true>;
  else kB = M.sym ? sc3_win_pass<W, NTW, true, false> : sc3_win_pass<W, NTW, false, false>;
#define DNM_LO(DM_, SY_) (lo_first ? (kern_t)sc3_lo_pass<A, NT, DM_, SY_, false> : (kern_t)sc3_lo_pass<A, NT, DM_, SY_, true>)
  switch (dm * 2 + (M.sym ? 1 : 0)) {
    case 0: kA = DNM_LO(0, false); break;
    case 1: kA = DNM_LO(0, true); break;
    case 2: kA = DNM_LO(1, false); break;
    case 3: kA = DNM_LO(1, true); break;
    case 4: kA = DNM_LO(2, false); break;
    default: kA = DNM_LO(2, true); break;
  }
#undef DNM_LO
  static std::map<const void *, bool> attr_done;
  for (auto kp : {std::make_pair((const void *)kA, ldsA), std::make_pair((const void *)kB, ldsB)})
    if (!attr_done[kp.first]) {
      DNM_HIP(hipFuncSetAttribute(kp.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kp.second));
      attr_done[kp.first] = true;
    }
  Sc3Call first = call, second = call;
  first.dot_out = nullptr;
  second.zinit = nullptr;
  second.zinit2 = nullptr;
  if (M.real) {
    // real vectors: the window pass is the complex kernel on the halved tables (pairs of entries as elements: every
    // offset it forms is even), the lo pass its own kernel on doubles
    DNM_CHECK(M.sym, "internal: real vectors need a real operator");
    using kern_r = void (*)(const Sc3Tab, const Sc3Op, const uint32_t *, const Sc3Call, const double *, double *);
    kern_r kR = nullptr;
    constexpr int NTR = sc3r_threads(NT), PPR = sc3r_pairs(A, NT);
    if (dm == 0) kR = lo_first ? (kern_r)sc3_lo_pass_r<A, NTR, PPR, 0, false> : (kern_r)sc3_lo_pass_r<A, NTR, PPR, 0, true>;
    else if (dm == 1) kR = lo_first ? (kern_r)sc3_lo_pass_r<A, NTR, PPR, 1, false> : (kern_r)sc3_lo_pass_r<A, NTR, PPR, 1, true>;
    else kR = lo_first ? (kern_r)sc3_lo_pass_r<A, NTR, PPR, 2, false> : (kern_r)sc3_lo_pass_r<A, NTR, PPR, 2, true>;
    constexpr size_t ldsR = (size_t)sc3_lo_cap_r(A, NT) * 8;
    if (!attr_done[(const void *)kR]) {
      DNM_HIP(hipFuncSetAttribute((const void *)kR, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsR));
      attr_done[(const void *)kR] = true;
    }
    Sc3Call firstw = phase == 0 ? first : second;
    firstw.row0 /= 2;
    firstw.win_start /= 2;
    if (phase == 0 || phase == 2)
      hipLaunchKernelGGL(kB, dim3((unsigned)M.permB.size()), dim3(NTW), ldsB, st, M.ly->dev_h, op, (const uint32_t *)M.d_permB,
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

size_t sc3_dot_partials(const Sc3Mat &M) { return M.permA.size() / 8; }

int launch_sc3(const Sc3Mat &M, const DevMsc &msc, const Sc3Call &call, const double *cached_diag, const void *xw,
               void *y, hipStream_t st, int phase) {
  DNM_CHECK(M.ly && M.ly->on_device, "layout tables are not on the device");
  DNM_CHECK(phase == 0 || (M.tiled && !call.dot_out), "internal: only the tiled passes split into a local and a remote part");
  if (M.tiled) {
    if (call.dot_out) DNM_HIP(hipMemsetAsync(call.dot_out, 0, sc3_dot_partials(M) * 3 * sizeof(double), st));
    if (M.graph) return launch_sc3g(M, call, cached_diag, xw, y, st, phase);
    if (M.ly->host.a == 14) {
      static const bool w1024 = [] { const char *e = knob("DNM_SC3_WIN_THREADS"); return e && atoi(e) == 1024; }();   // experiments
      if (w1024) return launch_two_pass<14, 10, 1024, 1024>(M, call, cached_diag, xw, y, st, phase);
      return launch_two_pass<14, 10, 1024, 512>(M, call, cached_diag, xw, y, st, phase);
    }
    // (256 threads for rows of at most 20 states: the small instance runs four rows per workgroup, so that the tests
    // at L = 11...24 cover the sub-group form of the lo pass)
    return launch_two_pass<6, 4, 256, 64>(M, call, cached_diag, xw, y, st, phase);
  }
  DNM_CHECK(!call.dot_out, "internal: the row kernel has no fused sums");
  if (M.rowsel.empty()) return 0;      // a rank that owns no rows
  hipLaunchKernelGGL(sc3_row_kernel, dim3((unsigned)M.rowsel.size()), dim3(SC3_ROW_NT), 0, st, M.ly->dev, msc,
                     (const uint32_t *)M.d_rowsel, call, cached_diag, (const c128 *)xw, (c128 *)y);
  DNM_HIP(hipGetLastError());
  return 0;
}

}  // namespace dnm
