// Prototype (round 3): SpinConserve(L,k) multiply in a three-field internal layout, two tiled passes.
//
//   state = [ T : t top bits | W : w window bits | Lo : a low bits ]
//   internal order: T ascending; inside a T block (kr = k - |T| ones left) classes cw = |W| ascending; inside a
//   class a dense matrix [wr = rank of W among the w-bit patterns with cw ones][lr = rank of Lo among the a-bit
//   patterns with kl = kr - cw ones], rows padded to a multiple of 8 amplitudes (128-byte lines).
//   The reference order (ascending state, bsubspace_impl.h:187-245) is the same T blocks with the rows of every
//   block in ascending W: the conversion permutes whole rows.
//
//   pass "lo"  : one workgroup per row (T, W): the a-1 bonds inside Lo from LDS; every other bond couples the row
//                to ONE other row at a uniform offset (a contiguous run), the Lo/W boundary bond for a contiguous
//                part of the row.
//   pass "win" : one workgroup per (T, cw, run of R = 16 << s amplitudes): all C(w, cw) window patterns x R
//                columns in LDS: the w-1 bonds inside W from LDS; T bonds and the W/T boundary bond at a uniform
//                offset.
//   Which pass gathers which of the non-LDS bonds is a bit mask (bondsA / bondsB).
//
// Build / run (GPU box):  hipcc --offload-arch=gfx950 -O3 tools/experiments/sc3_proto.hip -o /tmp/sc3_proto
//                         /tmp/sc3_proto L k [a w orderA orderB tInA accA reps ntA ntB nbA nbB t1 variant shb]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double2 c128;
typedef double d2v __attribute__((ext_vector_type(2)));

struct Sc3 {
  int32_t L, k, a, w, t;
  const int64_t *ibase;      // [1 << t]  internal offset of the T block (-1: no such block)
  const int64_t *icoff;      // [(a+w+1) * (w+1)]  offset of class cw inside a T block with kr ones: [kr * (w+1) + cw]
  int32_t nl[18], pitch[18];       // by kl: C(a, kl) and the padded row length
  int32_t lo_off[19];              // lo_pat group offsets by kl
  int32_t nw[14], w_off[15];       // by cw: C(w, cw), w_pat group offsets
  int32_t rs[14];                  // window pass: log2(R / 16) by cw
  const uint16_t *lo_pat, *w_pat, *w_rank;
  const int32_t *cbin;             // [17 * 17] C(n, j)
  const double *bond;              // [(L-1) * 4] up_re, up_im, dn_re, dn_im
  const double *diag;              // internal layout, or null
  const double *dlo;               // on-the-fly diagonal: the part that depends on Lo only, indexed like lo_pat
  const double *hfield;            // [L] field term of a site (+h for a zero bit, -h for a one)
  double zz;                       // ZZ coupling of every bond
  int32_t shb;                     // window pass: log2 of the shortest run (4: 256-byte runs, 3: 128-byte runs)
  uint64_t bondsA, bondsB;         // non-LDS bonds gathered by the lo pass / the window pass
};

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

constexpr int cbinom(int n, int k) {
  long long r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return (int)r;
}


// ---------------------------------------------------------------------------------------------------------
// lo pass: one workgroup per row (T, W)
// DIAGM: 0 no diagonal, 1 cached (8 B/row), 2 on the fly (table over Lo + per-row scalar + boundary bond)
// SYM: every bond coefficient is real and the same in both directions; EARLY: first gathers issued before the tile
// is waited for
template <int A, int NT, int NB, bool ACC, int DIAGM, bool SYM, bool EARLY>
__global__ void __launch_bounds__(NT, (2048 / NT) * NT / 256)
sc3_lo_pass(const Sc3 S, const uint32_t *__restrict__ perm, const c128 *__restrict__ x, c128 *__restrict__ y) {
  constexpr int MAXROWS = cbinom(A, A / 2);
  constexpr int RPT = (MAXROWS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c128 *xs = reinterpret_cast<c128 *>(smem);
  __shared__ int32_t cl[A * (A + 1)];
  const uint32_t e = perm[blockIdx.x];
  if (e == 0xffffffffu) return;
  const int lane = threadIdx.x & 63;
  const int w = S.w;
  const uint32_t T = e >> w, W = e & ((1u << w) - 1u);
  const int cw = __popc(W), kr = S.k - __popc(T), kl = kr - cw;
  const int nrows = S.nl[kl], p = S.pitch[kl];
  const int wr = S.w_rank[W];
  const int64_t tb = S.ibase[T];
  const int64_t base = tb + S.icoff[kr * (w + 1) + cw] + (int64_t)wr * p;

  uint32_t lowb[RPT];
  c128 xv[RPT];
  const uint16_t *__restrict__ pat = S.lo_pat + S.lo_off[kl];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = threadIdx.x + i * NT;
    lowb[i] = 0;
    xv[i] = make_double2(0.0, 0.0);
    if (r < nrows) {
      lowb[i] = pat[r];
      xv[i] = x[base + r];
    }
  }
  // bonds outside Lo, one per lane: l = 0 the Lo/W boundary, 1..w-1 inside W, w the W/T boundary, above inside T
  int act = 0, r0 = 0, r1 = nrows;
  int64_t delta = 0;
  double c0 = 0.0, c1 = 0.0;
  {
    const int b = A - 1 + lane;
    if (b < S.L - 1 && ((S.bondsA >> b) & 1ull)) {
      bool up = false;
      if (lane == 0) {
        const int cut = S.cbin[(A - 1) * 17 + kl];                   // rows below: top bit of Lo clear
        if (W & 1u) {                                                // the one comes down into Lo
          if (cut > 0) {
            act = 1; r0 = 0; r1 = cut; up = false;
            delta = tb + S.icoff[kr * (w + 1) + cw - 1] + (int64_t)S.w_rank[W & ~1u] * S.pitch[kl + 1] +
                    S.cbin[(A - 1) * 17 + kl + 1] - base;
          }
        } else if (cut < nrows) {                                    // the one goes up into W
          act = 1; r0 = cut; r1 = nrows; up = true;
          delta = tb + S.icoff[kr * (w + 1) + cw + 1] + (int64_t)S.w_rank[W | 1u] * S.pitch[kl - 1] - cut - base;
        }
      } else if (lane < w) {
        const int bw = lane - 1;
        const uint32_t pair = (W >> bw) & 3u;
        if (pair == 1u || pair == 2u) {
          act = 1; up = pair == 1u;
          delta = ((int64_t)S.w_rank[W ^ (3u << bw)] - wr) * p;
        }
      } else if (lane == w) {
        const uint32_t pair = ((W >> (w - 1)) & 1u) | ((T & 1u) << 1);
        if (pair == 1u) {
          act = 1; up = true;
          delta = S.ibase[T | 1u] + S.icoff[(kr - 1) * (w + 1) + cw - 1] +
                  (int64_t)S.w_rank[W & ~(1u << (w - 1))] * p - base;
        } else if (pair == 2u) {
          act = 1; up = false;
          delta = S.ibase[T & ~1u] + S.icoff[(kr + 1) * (w + 1) + cw + 1] +
                  (int64_t)S.w_rank[W | (1u << (w - 1))] * p - base;
        }
      } else {
        const int bt = lane - w - 1;
        const uint32_t pair = (T >> bt) & 3u;
        if (pair == 1u || pair == 2u) {
          act = 1; up = pair == 1u;
          delta = S.ibase[T ^ (3u << bt)] - tb;
        }
      }
      if (act) {
        c0 = S.bond[4 * b + (up ? 0 : 2)];
        c1 = S.bond[4 * b + (up ? 1 : 3)];
      }
    }
  }
  uint64_t hb = __ballot(act);

  for (int tt = threadIdx.x; tt < A * (A + 1); tt += NT) {
    const int lo = tt / (A + 1), o = tt % (A + 1);
    cl[tt] = S.cbin[lo * 17 + o];
  }
  double accr[RPT], acci[RPT];
  const c128 *__restrict__ pp[NB];
  double cr[NB], ci[NB];
  int q0[NB], q1[NB];
  c128 v[NB][RPT];
  auto setup = [&]() {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool have = hb != 0;
      const int m = have ? __ffsll((long long)hb) - 1 : 0;
      hb &= hb - 1;
      pp[j] = x + (base + rl_i64(delta, m));
      cr[j] = have ? rl_f64(c0, m) : 0.0;
      ci[j] = have ? rl_f64(c1, m) : 0.0;
      q0[j] = rl_i32(r0, m);
      q1[j] = have ? rl_i32(r1, m) : 0;
    }
  };
  auto issue = [&]() {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = threadIdx.x + i * NT;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        v[j][i] = make_double2(0.0, 0.0);
        if (r >= q0[j] && r < q1[j]) v[j][i] = pp[j][r];
      }
    }
  };
  auto consume = [&]() {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        accr[i] = fma(cr[j], v[j][i].x, accr[i]);
        acci[i] = fma(cr[j], v[j][i].y, acci[i]);
        if (!SYM) {
          accr[i] = fma(-ci[j], v[j][i].y, accr[i]);
          acci[i] = fma(ci[j], v[j][i].x, acci[i]);
        }
      }
    }
  };
  const bool any = hb != 0;
  if (EARLY && any) { setup(); issue(); }
  // on-the-fly diagonal: the part of the row (T, W): fields and bonds above Lo, one site per lane
  double dhi = 0.0;
  if (DIAGM == 2) {
    const uint64_t hi = ((uint64_t)T << w) | W;
    const int site = A + lane;
    double term = 0.0;
    if (site < S.L) {
      const uint32_t bit = (uint32_t)(hi >> lane) & 1u;
      term = bit ? -S.hfield[site] : S.hfield[site];
      if (site + 1 < S.L) term += (bit ^ ((uint32_t)(hi >> (lane + 1)) & 1u)) ? -S.zz : S.zz;
    }
    for (int off = 32; off > 0; off >>= 1) term += __shfl_xor(term, off, 64);
    dhi = rl_f64(term, 0);
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = threadIdx.x + i * NT;
    accr[i] = 0.0;
    acci[i] = 0.0;
    if (r < nrows) {
      xs[r] = xv[i];
      if (DIAGM == 1) {
        const double dg = __builtin_nontemporal_load(S.diag + base + r);
        accr[i] = dg * xv[i].x;
        acci[i] = dg * xv[i].y;
      } else if (DIAGM == 2) {
        // Lo part from the table (L2-resident), the bond across the boundary from the top bit of Lo and bit 0 of W
        const double dg = S.dlo[S.lo_off[kl] + r] + dhi + ((((lowb[i] >> (A - 1)) ^ W) & 1u) ? -S.zz : S.zz);
        accr[i] = dg * xv[i].x;
        acci[i] = dg * xv[i].y;
      }
    }
  }
  if (EARLY && any) consume();
  while (hb) { setup(); issue(); consume(); }
  __syncthreads();
  for (int lo = 0; lo < A - 1; ++lo) {
    const double ure = S.bond[4 * lo], uim = S.bond[4 * lo + 1], dre = S.bond[4 * lo + 2], dim_ = S.bond[4 * lo + 3];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = threadIdx.x + i * NT;
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
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = threadIdx.x + i * NT;
    if (r < p) {                                  // the padding of a row is written too (zeros)
      double ar = accr[i], ai = acci[i];
      if (ACC && r < nrows) {
        const c128 yo = load_nt(y + base + r);
        ar += yo.x;
        ai += yo.y;
      }
      store_nt(y + base + r, ar, ai);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// window pass: one workgroup per (T, cw, run of R = 16 << s columns): all window patterns of the class
template <int WB, int NT, int NB, bool ACC, int DIAGM, bool SYM, bool EARLY, int SHB = 4, bool PIPE = false>
__global__ void __launch_bounds__(NT, (2048 / NT) * NT / 256)
sc3_win_pass(const Sc3 S, const uint32_t *__restrict__ perm, const c128 *__restrict__ x, c128 *__restrict__ y) {
  constexpr int MAXE = cbinom(WB, WB / 2) * (1 << SHB);
  constexpr int RPT = (MAXE + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c128 *xs = reinterpret_cast<c128 *>(smem);
  __shared__ int32_t cl[WB * (WB + 1)];
  const uint32_t e = perm[blockIdx.x];
  if (e == 0xffffffffu) return;
  const int lane = threadIdx.x & 63;
  const uint32_t T = e >> 16;
  const int cw = (e >> 12) & 15, run = e & 0xfff;
  const int kr = S.k - __popc(T), kl = kr - cw;
  const int nwp = S.nw[cw], p = S.pitch[kl];
  const int sh = S.shb + S.rs[cw];
  const int lr0 = run << sh;
  const int ncols = min(1 << sh, p - lr0);
  const int64_t tb = S.ibase[T];
  const int64_t own = tb + S.icoff[kr * (WB + 1) + cw];
  const int64_t cbase = own + lr0;
  const int nent = nwp << sh;

  uint32_t wpat[RPT];
  int32_t off[RPT];          // offset of the entry from cbase, -1: not an entry
  c128 xv[RPT];
  const uint16_t *__restrict__ pat = S.w_pat + S.w_off[cw];
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
    if (b < S.L - 1 && ((S.bondsB >> b) & 1ull)) {
      bool up = false;
      if (lane == 0) {
        const int cut = S.cbin[(WB - 1) * 17 + cw];                // rows below: top bit of W clear
        if (T & 1u) {                                              // the one comes down into W
          if (cut > 0) {
            act = 1; r0 = 0; r1 = cut; up = false;
            delta = S.ibase[T & ~1u] + S.icoff[(kr + 1) * (WB + 1) + cw + 1] +
                    (int64_t)S.cbin[(WB - 1) * 17 + cw + 1] * p - own;
          }
        } else if (cut < nwp) {                                    // the one goes up into T
          act = 1; r0 = cut; r1 = nwp; up = true;
          delta = S.ibase[T | 1u] + S.icoff[(kr - 1) * (WB + 1) + cw - 1] - (int64_t)cut * p - own;
        }
      } else {
        const int bt = lane - 1;
        const uint32_t pair = (T >> bt) & 3u;
        if (pair == 1u || pair == 2u) {
          act = 1; up = pair == 1u;
          delta = S.ibase[T ^ (3u << bt)] - tb;
        }
      }
      if (act) {
        c0 = S.bond[4 * b + (up ? 0 : 2)];
        c1 = S.bond[4 * b + (up ? 1 : 3)];
      }
    }
  }
  uint64_t hb = __ballot(act);
  for (int tt = threadIdx.x; tt < WB * (WB + 1); tt += NT) {
    const int lo = tt / (WB + 1), o = tt % (WB + 1);
    cl[tt] = S.cbin[lo * 17 + o];
  }
  double accr[RPT], acci[RPT];
  const c128 *__restrict__ pp[NB];
  double cr[NB], ci[NB];
  int q0[NB], q1[NB];
  c128 v[NB][RPT];
  auto setup = [&]() {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool have = hb != 0;
      const int m = have ? __ffsll((long long)hb) - 1 : 0;
      hb &= hb - 1;
      pp[j] = x + (cbase + rl_i64(delta, m));
      cr[j] = have ? rl_f64(c0, m) : 0.0;
      ci[j] = have ? rl_f64(c1, m) : 0.0;
      q0[j] = rl_i32(r0, m);
      q1[j] = have ? rl_i32(r1, m) : 0;
    }
  };
  auto issue = [&]() {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int wrr = (int)(wpat[i] >> 16);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        v[j][i] = make_double2(0.0, 0.0);
        if (off[i] >= 0 && wrr >= q0[j] && wrr < q1[j]) v[j][i] = pp[j][off[i]];
      }
    }
  };
  auto consume = [&]() {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        accr[i] = fma(cr[j], v[j][i].x, accr[i]);
        acci[i] = fma(cr[j], v[j][i].y, acci[i]);
        if (!SYM) {
          accr[i] = fma(-ci[j], v[j][i].y, accr[i]);
          acci[i] = fma(ci[j], v[j][i].x, acci[i]);
        }
      }
    }
  };
  const bool any = hb != 0;
  if (EARLY && any) { setup(); issue(); }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int en = threadIdx.x + i * NT;
    accr[i] = 0.0;
    acci[i] = 0.0;
    if (en < nent) {
      xs[en] = xv[i];
      if (DIAGM == 1 && off[i] >= 0) {
        const double dg = __builtin_nontemporal_load(S.diag + cbase + off[i]);
        accr[i] = dg * xv[i].x;
        acci[i] = dg * xv[i].y;
      }
    }
  }
  // PIPE: the gathered bonds ride on the LDS phase -- one gather is in flight while two LDS bonds are worked off,
  // then it is consumed and the next one issued
  bool pend = false;
  if (PIPE) {
    if (any && !EARLY) { setup(); issue(); }
    pend = any;
  } else {
    if (EARLY && any) consume();
    while (hb) { setup(); issue(); consume(); }
  }
  __syncthreads();
  for (int lo = 0; lo < WB - 1; ++lo) {
    if (PIPE && pend && lo > 0 && (lo & 1) == 0) {
      consume();
      pend = false;
      if (hb) { setup(); issue(); pend = true; }
    }
    const int b = S.a + lo;
    const double ure = S.bond[4 * b], uim = S.bond[4 * b + 1], dre = S.bond[4 * b + 2], dim_ = S.bond[4 * b + 3];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int en = threadIdx.x + i * NT;
      const uint32_t pair = (wpat[i] >> lo) & 3u;
      if (off[i] >= 0 && (pair == 1u || pair == 2u)) {
        const bool up = pair == 1u;
        const int ord0 = __popc(wpat[i] & ((1u << lo) - 1u));
        const int d = cl[lo * (WB + 1) + ord0] << sh;
        const c128 xp = xs[up ? en + d : en - d];
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
  if (PIPE) {
    if (pend) consume();
    while (hb) { setup(); issue(); consume(); }
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    if (off[i] >= 0) {
      double ar = accr[i], ai = acci[i];
      if (ACC) {
        const c128 yo = load_nt(y + cbase + off[i]);
        ar += yo.x;
        ai += yo.y;
      }
      store_nt(y + cbase + off[i], ar, ai);
    }
  }
}

// timing runs: x and a stand-in diagonal filled on the device, row by row (padding zero)
__global__ void __launch_bounds__(256) sc3_fill(const Sc3 S, const uint32_t *__restrict__ perm, c128 *__restrict__ x,
                                                double *__restrict__ diag) {
  const uint32_t e = perm[blockIdx.x];
  if (e == 0xffffffffu) return;
  const int w = S.w;
  const uint32_t T = e >> w, W = e & ((1u << w) - 1u);
  const int cw = __popc(W), kr = S.k - __popc(T), kl = kr - cw;
  const int nrows = S.nl[kl], p = S.pitch[kl];
  const int64_t base = S.ibase[T] + S.icoff[kr * (w + 1) + cw] + (int64_t)S.w_rank[W] * p;
  for (int r = threadIdx.x; r < p; r += 256) {
    const uint64_t z = ((uint64_t)(base + r) * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(base + r) >> 17);
    const bool live = r < nrows;
    x[base + r] = live ? make_double2((double)(z & 0xffff) / 65536.0 - 0.5, (double)((z >> 16) & 0xffff) / 65536.0 - 0.5)
                       : make_double2(0.0, 0.0);
    diag[base + r] = live ? (double)((z >> 32) & 0xff) / 64.0 - 2.0 : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------------------
static int64_t binom(int n, int k) {
  if (k < 0 || k > n) return 0;
  long double r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return (int64_t)llroundl(r);
}

template <class T_>
static T_ *upload(const std::vector<T_> &v) {
  T_ *d;
  CK(hipMalloc(&d, std::max<size_t>(1, v.size()) * sizeof(T_)));
  CK(hipMemcpy(d, v.data(), v.size() * sizeof(T_), hipMemcpyHostToDevice));
  return d;
}

// deal groups of workgroups to the 8 XCDs (workgroup b runs on XCD b % 8): stream s takes groups s, s+8, ...
static std::vector<uint32_t> deal(const std::vector<std::vector<uint32_t>> &groups, bool use_groups) {
  std::vector<uint32_t> out;
  if (!use_groups) {
    for (auto &g : groups) out.insert(out.end(), g.begin(), g.end());
    return out;
  }
  std::vector<std::vector<uint32_t>> st(8);
  // keep the streams level: the next group goes to the shortest stream
  for (auto &g : groups) {
    int best = 0;
    for (int s = 1; s < 8; ++s) if (st[s].size() < st[best].size()) best = s;
    st[best].insert(st[best].end(), g.begin(), g.end());
  }
  size_t n = 0;
  for (auto &s : st) n = std::max(n, s.size());
  out.assign(8 * n, 0xffffffffu);
  for (int s = 0; s < 8; ++s)
    for (size_t i = 0; i < st[s].size(); ++i) out[8 * i + s] = st[s][i];
  return out;
}

int main(int argc, char **argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 26, k = argc > 2 ? atoi(argv[2]) : L / 2;
  const int a = argc > 3 ? atoi(argv[3]) : 14, w = argc > 4 ? atoi(argv[4]) : 10;
  const int orderA = argc > 5 ? atoi(argv[5]) : 1, orderB = argc > 6 ? atoi(argv[6]) : 1;
  const int tInA = argc > 7 ? atoi(argv[7]) : 0;     // 1: T bonds and the W/T boundary gathered by the lo pass
  const int accA = argc > 8 ? atoi(argv[8]) : 1;     // 1: window pass first, lo pass accumulates (+ diagonal)
  const int reps = argc > 9 ? atoi(argv[9]) : 5;
  const int t = L - a - w;
  // t1 (argv[14]): split of the T bonds between the passes when tInA == 2: the window pass gathers the W/T boundary
  // and the bonds inside the low t1 bits of T, the lo pass the bonds above (and the Lo/W boundary)
  const int t1 = argc > 14 ? atoi(argv[14]) : t / 2;
  const int shb = argc > 16 ? atoi(argv[16]) : 4;
  if (t < 1 || t > 15 || a != 14 || w != 10) { printf("prototype instances: a=14 w=10, 1 <= t <= 15\n"); return 1; }

  // ---- tables
  Sc3 S;
  memset(&S, 0, sizeof S);
  S.L = L; S.k = k; S.a = a; S.w = w; S.t = t;
  S.shb = shb;
  std::vector<int32_t> cbin(17 * 17);
  for (int n = 0; n < 17; ++n) for (int j = 0; j < 17; ++j) cbin[n * 17 + j] = (int32_t)binom(n, j);
  std::vector<uint16_t> lo_pat, w_pat, w_rank(1 << w);
  for (int j = 0; j <= a; ++j) {
    S.lo_off[j] = (int32_t)lo_pat.size();
    S.nl[j] = (int32_t)binom(a, j);
    S.pitch[j] = (S.nl[j] + 7) / 8 * 8;
    for (uint32_t v = 0; v < (1u << a); ++v) if (__builtin_popcount(v) == j) lo_pat.push_back((uint16_t)v);
  }
  S.lo_off[a + 1] = (int32_t)lo_pat.size();
  for (int j = 0; j <= w; ++j) {
    S.w_off[j] = (int32_t)w_pat.size();
    S.nw[j] = (int32_t)binom(w, j);
    int r = 0;
    for (uint32_t v = 0; v < (1u << w); ++v) if (__builtin_popcount(v) == j) { w_rank[v] = (uint16_t)r++; w_pat.push_back((uint16_t)v); }
    int s = 0;
    while ((S.nw[j] << (s + 1)) <= (int)binom(w, w / 2)) ++s;        // R = 16 << s keeps nw * R within the largest tile
    S.rs[j] = s;
  }
  S.w_off[w + 1] = (int32_t)w_pat.size();
  std::vector<int64_t> icoff((a + w + 1) * (w + 1), 0), isize(a + w + 1, 0);
  for (int kr = 0; kr <= a + w; ++kr) {
    int64_t o = 0;
    for (int cw = 0; cw <= w; ++cw) {
      icoff[kr * (w + 1) + cw] = o;
      const int kl = kr - cw;
      if (kl >= 0 && kl <= a) o += binom(w, cw) * S.pitch[kl];
    }
    isize[kr] = o;
  }
  std::vector<int64_t> ibase(1 << t, -1);
  int64_t nint = 0, dim = 0;
  for (uint32_t T = 0; T < (1u << t); ++T) {
    const int kr = k - __builtin_popcount(T);
    if (kr < 0 || kr > a + w) continue;
    ibase[T] = nint;
    nint += isize[kr];
    dim += binom(a + w, kr);
  }
  printf("SpinConserve(%d,%d): dim %lld, internal length %lld (+%.3f%%), fields t=%d w=%d a=%d\n", L, k, (long long)dim,
         (long long)nint, 100.0 * (nint - dim) / dim, t, w, a);

  // chain Hamiltonian: 0.25 (XX + YY + ZZ) + random fields: flip-flop amplitude 0.5 per bond
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(-3.0, 3.0);
  std::vector<double> h(L), bond(4 * (L - 1), 0.0);
  for (int i = 0; i < L; ++i) h[i] = 0.5 * U(rng);
  for (int b = 0; b < L - 1; ++b) { bond[4 * b] = 0.5; bond[4 * b + 2] = 0.5; }
  auto diag_of = [&](uint64_t s) {
    double d = 0.0;
    for (int b = 0; b < L - 1; ++b) d += (((s >> b) ^ (s >> (b + 1))) & 1) ? -0.25 : 0.25;
    for (int i = 0; i < L; ++i) d += ((s >> i) & 1) ? -h[i] : h[i];
    return d;
  };

  // which pass gathers which bond
  uint64_t bA = 0, bB = 0;
  bA |= 1ull << (a - 1);                                         // Lo/W boundary: always the lo pass
  for (int b = a + w - 1; b < L - 1; ++b) {
    const int bt = b - (a + w);                       // -1: the W/T boundary; bt: bond between T bits bt, bt+1
    const bool toA = tInA == 2 ? (bt >= t1 - 1 && bt >= 0) : tInA != 0;
    (toA ? bA : bB) |= 1ull << b;
  }
  S.bondsA = bA; S.bondsB = bB;

  // ---- block lists
  std::vector<std::vector<uint32_t>> gA, gB;
  {
    // lo pass, order 1: groups = (kt, cw, wr), all T of the class (T-bond partners meet in one XCD's L2);
    // order 2: groups = (T, W >> 1) pairs; order 0: internal order
    std::vector<std::vector<uint32_t>> Tby(t + 1);
    for (uint32_t T = 0; T < (1u << t); ++T) if (ibase[T] >= 0) Tby[__builtin_popcount(T)].push_back(T);
    if (orderA == 3) {
      // small groups: T's that share the low t1-1 bits and the popcount (closed under the T bonds of the lo pass),
      // times the pair W, W^1 (the Lo/W boundary)
      const uint32_t lowm = t1 >= 1 ? (1u << (t1 - 1)) - 1u : 0u;
      for (int kt = 0; kt <= t; ++kt) {
        if (Tby[kt].empty()) continue;
        for (uint32_t tl = 0; tl <= lowm; ++tl) {
          std::vector<uint32_t> Ts;
          for (uint32_t T : Tby[kt]) if ((T & lowm) == tl) Ts.push_back(T);
          if (Ts.empty()) continue;
          const int kr = k - kt;
          for (uint32_t W2 = 0; W2 < (1u << (w - 1)); ++W2) {
            std::vector<uint32_t> g;
            for (uint32_t W : {2 * W2, 2 * W2 + 1}) {
              const int kl = kr - __builtin_popcount(W);
              if (kl < 0 || kl > a) continue;
              for (uint32_t T : Ts) g.push_back((T << w) | W);
            }
            if (!g.empty()) gA.push_back(g);
          }
        }
      }
    } else if (orderA == 1) {
      for (int kt = 0; kt <= t; ++kt) {
        if (Tby[kt].empty()) continue;
        const int kr = k - kt;
        for (int cw = 0; cw <= w; ++cw) {
          const int kl = kr - cw;
          if (kl < 0 || kl > a) continue;
          for (int wr = 0; wr < S.nw[cw]; ++wr) {
            std::vector<uint32_t> g;
            for (uint32_t T : Tby[kt]) g.push_back((T << w) | w_pat[S.w_off[cw] + wr]);
            gA.push_back(g);
          }
        }
      }
    } else {
      for (uint32_t T = 0; T < (1u << t); ++T) {
        if (ibase[T] < 0) continue;
        const int kr = k - __builtin_popcount(T);
        if (orderA == 2) {
          for (uint32_t W2 = 0; W2 < (1u << (w - 1)); ++W2) {
            std::vector<uint32_t> g;
            for (uint32_t W : {2 * W2, 2 * W2 + 1}) {
              const int kl = kr - __builtin_popcount(W);
              if (kl >= 0 && kl <= a) g.push_back((T << w) | W);
            }
            if (!g.empty()) gA.push_back(g);
          }
        } else {
          for (int cw = 0; cw <= w; ++cw) {
            const int kl = kr - cw;
            if (kl < 0 || kl > a) continue;
            std::vector<uint32_t> g;
            for (int wr = 0; wr < S.nw[cw]; ++wr) g.push_back((T << w) | w_pat[S.w_off[cw] + wr]);
            gA.push_back(g);
          }
        }
      }
    }
    // window pass, order 1: groups = (kt, cw, run), all T of the class; order 0: internal order
    if (orderB == 3) {
      // small groups: T's that share the bits above t1 and the popcount of the low t1 bits
      for (uint32_t T2 = 0; T2 < (1u << (t - t1)); ++T2) {
        for (int k1 = 0; k1 <= t1; ++k1) {
          std::vector<uint32_t> Ts;
          for (uint32_t T1 = 0; T1 < (1u << t1); ++T1)
            if (__builtin_popcount(T1) == k1 && ibase[(T2 << t1) | T1] >= 0) Ts.push_back((T2 << t1) | T1);
          if (Ts.empty()) continue;
          const int kr = k - __builtin_popcount(Ts[0]);
          for (int cw = 0; cw <= w; ++cw) {
            const int kl = kr - cw;
            if (kl < 0 || kl > a) continue;
            const int R = (1 << S.shb) << S.rs[cw], nrun = (S.pitch[kl] + R - 1) / R;
            for (int run = 0; run < nrun; ++run) {
              std::vector<uint32_t> g;
              for (uint32_t T : Ts) g.push_back((T << 16) | (cw << 12) | run);
              gB.push_back(g);
            }
          }
        }
      }
    } else if (orderB == 1) {
      for (int kt = 0; kt <= t; ++kt) {
        if (Tby[kt].empty()) continue;
        const int kr = k - kt;
        for (int cw = 0; cw <= w; ++cw) {
          const int kl = kr - cw;
          if (kl < 0 || kl > a) continue;
          const int R = (1 << S.shb) << S.rs[cw], nrun = (S.pitch[kl] + R - 1) / R;
          for (int run = 0; run < nrun; ++run) {
            std::vector<uint32_t> g;
            for (uint32_t T : Tby[kt]) g.push_back((T << 16) | (cw << 12) | run);
            gB.push_back(g);
          }
        }
      }
    } else {
      for (uint32_t T = 0; T < (1u << t); ++T) {
        if (ibase[T] < 0) continue;
        const int kr = k - __builtin_popcount(T);
        for (int cw = 0; cw <= w; ++cw) {
          const int kl = kr - cw;
          if (kl < 0 || kl > a) continue;
          const int R = (1 << S.shb) << S.rs[cw], nrun = (S.pitch[kl] + R - 1) / R;
          std::vector<uint32_t> g;
          for (int run = 0; run < nrun; ++run) g.push_back((T << 16) | (cw << 12) | run);
          gB.push_back(g);
        }
      }
    }
  }
  const std::vector<uint32_t> permA = deal(gA, orderA != 0), permB = deal(gB, orderB != 0);
  printf("lo pass: %zu workgroups (order %d), window pass: %zu (order %d); T bonds gathered by the %s pass (t1=%d); %s pass accumulates\n",
         permA.size(), orderA, permB.size(), orderB, tInA == 2 ? "lo/window" : tInA ? "lo" : "window", t1, accA ? "lo" : "window");

  // ---- vectors (internal layout), x random on the rows, zero in the padding
  const bool check = dim <= (int64_t)60e6;
  std::vector<c128> hx;
  std::vector<double> hdiag;
  std::vector<uint64_t> state_of;          // internal position -> state (check only), ~0 for padding
  c128 *dx, *dy;
  double *ddiag;
  CK(hipMalloc(&dx, nint * sizeof(c128)));
  CK(hipMalloc(&dy, nint * sizeof(c128)));
  CK(hipMalloc(&ddiag, nint * sizeof(double)));
  auto pos_of = [&](uint64_t s) -> int64_t {
    const uint32_t T = (uint32_t)(s >> (a + w)), W = (uint32_t)(s >> a) & ((1u << w) - 1), Lo = (uint32_t)s & ((1u << a) - 1);
    const int cw = __builtin_popcount(W), kr = k - __builtin_popcount(T), kl = kr - cw;
    // rank of Lo among the a-bit patterns with kl ones (colex)
    int64_t lr = 0;
    int o = 0;
    for (int pbit = 0; pbit < a; ++pbit) if ((Lo >> pbit) & 1) { ++o; lr += binom(pbit, o); }
    return ibase[T] + icoff[kr * (w + 1) + cw] + (int64_t)w_rank[W] * S.pitch[kl] + lr;
  };
  if (check) {
    hx.assign(nint, make_double2(0.0, 0.0));
    hdiag.assign(nint, 0.0);
    state_of.assign(nint, ~0ull);
    std::mt19937_64 r2(7);
    std::normal_distribution<double> N01(0.0, 1.0);
    for (uint32_t T = 0; T < (1u << t); ++T) {
      if (ibase[T] < 0) continue;
      const int kr = k - __builtin_popcount(T);
      for (uint32_t W = 0; W < (1u << w); ++W) {
        const int cw = __builtin_popcount(W), kl = kr - cw;
        if (kl < 0 || kl > a) continue;
        const int64_t rb = ibase[T] + icoff[kr * (w + 1) + cw] + (int64_t)w_rank[W] * S.pitch[kl];
        const uint64_t hi = ((uint64_t)T << (a + w)) | ((uint64_t)W << a);
        for (int r = 0; r < S.nl[kl]; ++r) {
          const uint64_t s = hi | lo_pat[S.lo_off[kl] + r];
          hx[rb + r] = make_double2(N01(r2), N01(r2));
          state_of[rb + r] = s;
          hdiag[rb + r] = diag_of(s);
        }
      }
    }
    CK(hipMemcpy(dx, hx.data(), nint * sizeof(c128), hipMemcpyHostToDevice));
    CK(hipMemcpy(ddiag, hdiag.data(), nint * sizeof(double), hipMemcpyHostToDevice));
  }
  CK(hipMemset(dy, 0xff, nint * sizeof(c128)));
  S.ibase = upload(ibase); S.icoff = upload(icoff);
  S.lo_pat = upload(lo_pat); S.w_pat = upload(w_pat); S.w_rank = upload(w_rank);
  S.cbin = upload(cbin); S.bond = upload(bond); S.diag = ddiag;
  {
    std::vector<double> dlo(lo_pat.size());
    for (size_t i = 0; i < lo_pat.size(); ++i) {
      const uint32_t v = lo_pat[i];
      double d = 0.0;
      for (int b = 0; b + 1 < a; ++b) d += (((v >> b) ^ (v >> (b + 1))) & 1) ? -0.25 : 0.25;
      for (int i2 = 0; i2 < a; ++i2) d += ((v >> i2) & 1) ? -h[i2] : h[i2];
      dlo[i] = d;
    }
    S.dlo = upload(dlo);
    S.hfield = upload(h);
    S.zz = 0.25;
  }
  uint32_t *dpA = upload(permA), *dpB = upload(permB);
  if (!check) {
    hipLaunchKernelGGL(sc3_fill, dim3((unsigned)permA.size()), dim3(256), 0, 0, S, dpA, dx, ddiag);
    CK(hipDeviceSynchronize());
  }

  const int ntA = argc > 10 ? atoi(argv[10]) : 512, ntB = argc > 11 ? atoi(argv[11]) : 512;
  const int nbA = argc > 12 ? atoi(argv[12]) : 2, nbB = argc > 13 ? atoi(argv[13]) : 2;
  const size_t ldsA = (size_t)cbinom(14, 7) * 16, ldsB = (size_t)cbinom(10, 5) * 16 * ((size_t)1 << S.shb);
  using kern_t = void (*)(const Sc3, const uint32_t *, const c128 *, c128 *);
  kern_t kA_acc = nullptr, kA_first = nullptr, kB_acc = nullptr, kB_first = nullptr;
  // variant (argv[15]): bit 0 = on-the-fly diagonal instead of the cached one, bit 1 = SYM, bit 2 = EARLY
  const int variant = argc > 15 ? atoi(argv[15]) : 0;
#define PICKV(NT_, NB_, DM_, SY_, EA_)                                                                             \
  if (ntA == NT_ && nbA == NB_ && variant == ((DM_ == 2 ? 1 : 0) | (SY_ ? 2 : 0) | (EA_ ? 4 : 0))) {                 \
    kA_acc = sc3_lo_pass<14, NT_, NB_, true, DM_, SY_, EA_>;                                                        \
    kA_first = sc3_lo_pass<14, NT_, NB_, false, (DM_ == 2 ? 2 : 0), SY_, EA_>;                                                       \
  }                                                                                                                \
  if (ntB == NT_ && nbB == NB_ && variant == ((DM_ == 2 ? 1 : 0) | (SY_ ? 2 : 0) | (EA_ ? 4 : 0))) {                 \
    kB_acc = sc3_win_pass<10, NT_, NB_, true, (DM_ == 2 ? 0 : 1), SY_, EA_>;                                        \
    kB_first = sc3_win_pass<10, NT_, NB_, false, 0, SY_, EA_>;                                                      \
  }
#define PICKALL(NT_, NB_) PICKV(NT_, NB_, 1, false, false) PICKV(NT_, NB_, 2, false, false) PICKV(NT_, NB_, 1, true, false) \
  PICKV(NT_, NB_, 2, true, false) PICKV(NT_, NB_, 1, false, true) PICKV(NT_, NB_, 2, false, true) PICKV(NT_, NB_, 1, true, true) \
  PICKV(NT_, NB_, 2, true, true)
  PICKALL(512, 1) PICKALL(512, 2) PICKALL(1024, 1) PICKALL(1024, 2)
  if (variant == 19) {     // gathers pipelined with the LDS phase (window pass), on-the-fly diagonal, symmetric bonds
    kA_acc = sc3_lo_pass<14, 1024, 1, true, 2, true, false>;
    kA_first = sc3_lo_pass<14, 1024, 1, false, 2, true, false>;
    kB_acc = sc3_win_pass<10, 1024, 1, true, 0, true, true, 4, true>;
    kB_first = sc3_win_pass<10, 1024, 1, false, 0, true, true, 4, true>;
  }
  if (shb == 3) {          // 128-byte runs: 31.5 KB tiles, 512 threads x 4 entries, four workgroups per CU
    const bool sy = (variant & 2) != 0;
    kB_acc = sy ? sc3_win_pass<10, 512, 1, true, 0, true, false, 3> : sc3_win_pass<10, 512, 1, true, 0, false, false, 3>;
    kB_first = sy ? sc3_win_pass<10, 512, 1, false, 0, true, false, 3> : sc3_win_pass<10, 512, 1, false, 0, false, false, 3>;
  }
  if (!kA_acc || !kB_acc) { printf("no such kernel instance\n"); return 1; }
  CK(hipFuncSetAttribute((const void *)kA_acc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));
  CK(hipFuncSetAttribute((const void *)kA_first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));
  CK(hipFuncSetAttribute((const void *)kB_acc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
  CK(hipFuncSetAttribute((const void *)kB_first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
  auto passA = [&](bool acc) {
    hipLaunchKernelGGL(acc ? kA_acc : kA_first, dim3((unsigned)permA.size()), dim3(ntA), ldsA, 0, S, dpA, dx, dy);
  };
  auto passB = [&](bool acc) {
    hipLaunchKernelGGL(acc ? kB_acc : kB_first, dim3((unsigned)permB.size()), dim3(ntB), ldsB, 0, S, dpB, dx, dy);
  };
  auto multiply = [&]() {
    if (accA) { passB(false); passA(true); }
    else { passA(false); passB(true); }
  };
  multiply();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());

  if (check) {
    std::vector<c128> hy(nint);
    CK(hipMemcpy(hy.data(), dy, nint * sizeof(c128), hipMemcpyDeviceToHost));
    double maxerr = 0.0, maxpad = 0.0, nrm = 0.0;
    for (int64_t i = 0; i < nint; ++i) {
      const uint64_t s = state_of[i];
      if (s == ~0ull) { maxpad = std::max(maxpad, std::fabs(hy[i].x) + std::fabs(hy[i].y)); continue; }
      double re = hdiag[i] * hx[i].x, im = hdiag[i] * hx[i].y;
      for (int b = 0; b < L - 1; ++b) {
        if ((((s >> b) ^ (s >> (b + 1))) & 1) == 0) continue;
        const c128 xp = hx[pos_of(s ^ (3ull << b))];
        re += 0.5 * xp.x;
        im += 0.5 * xp.y;
      }
      maxerr = std::max(maxerr, std::max(std::fabs(re - hy[i].x), std::fabs(im - hy[i].y)));
      nrm = std::max(nrm, std::fabs(re));
    }
    printf("check against the definition: max |err| %.3e (max |y| %.2f), padding max %.3e  %s\n", maxerr, nrm, maxpad,
           (maxerr < 1e-12 && maxpad == 0.0) ? "OK" : "FAIL");
  }

  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float tA = 0, tB = 0;
  for (int r = 0; r < reps; ++r) {
    float m1, m2;
    CK(hipEventRecord(e0, 0));
    if (accA) passB(false); else passA(false);
    CK(hipEventRecord(e1, 0));
    if (accA) passA(true); else passB(true);
    CK(hipEventRecord(e2, 0));
    CK(hipEventSynchronize(e2));
    CK(hipEventElapsedTime(&m1, e0, e1));
    CK(hipEventElapsedTime(&m2, e1, e2));
    if (accA) { tB += m1; tA += m2; } else { tA += m1; tB += m2; }
  }
  tA /= reps; tB /= reps;
  printf("L=%d k=%d a=%d w=%d orderA=%d orderB=%d tInA=%d accA=%d ntA=%d ntB=%d nbA=%d nbB=%d t1=%d variant=%d: lo pass %.3f ms, window pass %.3f ms, multiply %.3f ms = %.2f Gamp/s, %.1f GB/s at 40 B/row\n",
         L, k, a, w, orderA, orderB, tInA, accA, ntA, ntB, nbA, nbB, t1, variant, tA, tB, tA + tB, dim / (tA + tB) / 1e6, 40.0 * dim / (tA + tB) / 1e6);
  return 0;
}
