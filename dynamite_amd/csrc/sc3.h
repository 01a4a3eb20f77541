// Three-field internal layout of SpinConserve(L, k) state vectors ("sc3") and the tables of its kernels.
//
//   state = [ T : t top bits | W : w window bits | Lo : a low bits ],  t = L - a - w >= 1
//
// The reference orders the basis by ascending state (bsubspace_impl.h:187-245): blocks of equal T, inside a block
// ascending W, inside that ascending Lo.  Rows (T, W) -- the C(a, kl) states that share T and W, kl = k - |T| - |W| --
// are contiguous in that order.  The internal layout keeps the T blocks where they are and, inside a block,
//   * groups the rows by cw = |W| (classes), ascending,
//   * orders the rows of a class by the rank wr of W among the w-bit patterns with cw ones,
//   * pads every row to a multiple of 8 amplitudes (128-byte lines); padding holds zeros.
// A class is then a dense matrix [wr][lr] with a fixed pitch: a bond inside Lo couples columns of one row, a bond
// inside W couples rows of one class at the same column, every other chain bond couples a row to ONE other row at a
// uniform offset.  Two tiled passes cover a nearest-neighbour chain (sc3_kernels.hip):
//   lo pass     one workgroup per row: the a-1 bonds inside Lo from LDS, the Lo/W boundary bond gathered;
//   window pass one workgroup per (T, cw, run of columns): all C(w, cw) rows of the class x R columns in LDS: the w-1
//               bonds inside W from LDS, the W/T boundary and the bonds inside T gathered at uniform offsets.
// Nothing like this exists in the reference (a PETSc Vec is opaque there as well); the maps to and from the
// reference order are dnm_vec_layout_copy / dnm_vec_layout_positions.
#pragma once

#include <array>
#include <cstdint>
#include <utility>
#include <vector>

#include <hip/hip_runtime.h>

#include "kernels.h"

namespace dnm {

constexpr int SC3_MAXA = 16, SC3_MAXW = 12;

// vec_swizzle code of a SpinConserve subspace: a | w << 8 (0: reference order)
static inline int sc3_code(int a, int w) { return a | (w << 8); }
static inline int sc3_code_a(int code) { return code & 0xff; }
static inline int sc3_code_w(int code) { return (code >> 8) & 0xff; }
// Order of the T blocks in memory (bits 16-19 of the code).  0: ascending T -- the reference's order, so that a rank's
// contiguous share of the layout is a contiguous range of the reference order too.  1 (round 6; vectors that live inside a
// solver): by (popcount(T >> 1), popcount of T's upper half, T >> 1, T & 1).  The two blocks that differ in T's lowest bit
// -- partners under the W/T boundary bond -- lie side by side; a chain's bonds inside T >> 1 keep the first key; only ONE
// bond -- between T's two lowest bits -- changes it.  A partition into contiguous ranges of this order cuts far fewer hops
// than ranges of ascending T, which cut every bond among the top log2(ranks) + 1 spins: SpinConserve(36,18) on 8 ranks,
// busiest rank: 14.3 GiB received per multiply instead of 42.2 (complex128), mean 11.3 instead of 25.5.  A rank's share is then NO range of the reference
// order: the maps to and from it (dnm_vec_layout_copy / _positions) serve whole vectors only.
static inline int sc3_code_order(int code) { return (code >> 16) & 0xf; }

struct Sc3Tab {
  int32_t L, k, a, w, t;
  int64_t nint;                    // length of a vector in this layout (rows + padding)
  const int64_t *ibase;            // [1 << t]  internal offset of the T block (-1: the block is empty)
  const int64_t *nbase;            // [1 << t]  reference index of the first state of the T block
  const int64_t *icoff;            // [(a+w+1) * (w+1)]  offset of class cw inside a T block with kr ones left
  const int64_t *ncoff;            // [(a+w+1) * (1 << w)]  reference offset of row W inside such a block
  int32_t nl[SC3_MAXA + 2], pitch[SC3_MAXA + 2];   // by kl: C(a, kl), padded row length
  int32_t lo_off[SC3_MAXA + 3];                    // lo_pat group offsets by kl
  int32_t nw[SC3_MAXW + 2], w_off[SC3_MAXW + 3];   // by cw: C(w, cw), w_pat group offsets
  int32_t rs[SC3_MAXW + 2];                        // window pass: log2(R / 16) by cw
  const uint16_t *lo_pat, *w_pat;  // patterns grouped by popcount, ascending inside a group
  const uint16_t *lo_rank, *w_rank;   // [1 << a], [1 << w]: rank of a pattern inside its group
  // Partner table of the window pass's LDS bonds (one lookup per (row, bond) instead of a pair test, an ordinal
  // popcount and a binomial): w_nb[2 * (w_off[cw] + wr)]: byte b of the 16 is the rank of the W pattern that bond b
  // (inside W) couples pattern wr to, or nw[cw] -- the zero row behind the tile -- when the two spins are equal.
  const uint64_t *w_nb;
  const int32_t *cbin;             // [17 * 17] C(n, j)
  // lo_rank in two halves (operators on any bond graph, sc3g_kernels.hip: 2.3 KB that a workgroup keeps in LDS where
  // lo_rank itself is 32 KB): with h = a / 2, lo = p & (2^h - 1), hi = p >> h:
  //   lo_rank[p] = lo_rlo[lo] + lo_rhi[hi * (h + 1) + popcount(lo)]
  const uint16_t *lo_rlo, *lo_rhi;
  const int64_t *nck;              // [(k+1) * (L+1)] C(LL, kk) at kk * (L+1) + LL: the reference's unranking table
};

// position of a state of the subspace in the internal layout
__host__ __device__ __forceinline__ int64_t sc3_pos(uint64_t state, const Sc3Tab &S) {
  const uint32_t T = (uint32_t)(state >> (S.a + S.w));
  const uint32_t W = (uint32_t)(state >> S.a) & ((1u << S.w) - 1u);
  const uint32_t Lo = (uint32_t)state & ((1u << S.a) - 1u);
#if defined(__HIP_DEVICE_COMPILE__)
  const int cw = __popc(W), kr = S.k - __popc(T);
#else
  const int cw = __builtin_popcount(W), kr = S.k - __builtin_popcount(T);
#endif
  return S.ibase[T] + S.icoff[kr * (S.w + 1) + cw] + (int64_t)S.w_rank[W] * S.pitch[kr - cw] + S.lo_rank[Lo];
}

// Site relabelling on top of the layout (dnm_subspace::site_perm): spin i of the reference's labelling is bit
// to_int[i] of the states the layout orders.  SpinConserve is invariant under it; the layout's tables are too.  The
// operator is rewritten into the layout's labelling when it is built (dnm_mat_create), vectors pass through it
// wherever they meet the reference order (dnm_vec_layout_copy / _positions / _set_random).  One rank only.
struct Sc3Perm {
  uint8_t to_int[64];              // reference spin -> layout bit
  uint8_t to_ref[64];              // layout bit -> reference spin
  int32_t L = 0;
  int32_t on = 0;                  // 0: identity
};
// from the descriptor's array (null: identity); false if it is not a permutation of 0 .. L-1
bool sc3_perm_make(const int8_t *site_perm, int L, Sc3Perm *out);
__host__ __device__ __forceinline__ uint64_t sc3_permute(uint64_t v, const uint8_t *map, int L) {
  uint64_t r = 0;
  for (int b = 0; b < L; ++b) r |= ((v >> b) & 1ull) << map[b];
  return r;
}
// Choose the relabelling for an operator: the assignment of its spins to the fields [T | W | Lo] that leaves the
// fewest (cheapest) hops between fields -- hops inside Lo or W come from LDS, the others are gathered (weights in
// sc3_perm.cpp).  masks: the operator's distinct masks; fix_top: spin L-1 keeps bit L-1 (XParity).  The identity is
// kept when nothing found is cheaper.  counts[6]: hops of the result inside Lo, inside W, inside T, Lo-W, Lo-T, W-T.
void sc3_choose_perm(int L, int a, int w, int64_t nmasks, const int64_t *masks, bool fix_top, int8_t *site_perm,
                     int32_t *counts);

// Host side: owns the tables (and, optionally, their device mirrors)
struct Sc3Layout {
  Sc3Tab host{}, dev{};
  // The same tables with every position halved (ibase, icoff, pitch, nint): a REAL vector of this layout read as
  // complex128 elements of two adjacent entries.  The window pass never looks inside a row, so on real vectors it is
  // the complex kernel run on these tables (rows of pitch / 2 elements).
  Sc3Tab host_h{}, dev_h{};
  std::vector<int64_t> ibase_h, icoff_h;
  void *d_ibase_h = nullptr, *d_icoff_h = nullptr;
  int64_t dim = 0;                 // C(L, k)
  bool on_device = false;
  std::vector<int64_t> ibase, nbase, icoff, ncoff, nck;
  std::vector<uint16_t> lo_pat, w_pat, lo_rank, w_rank;
  std::vector<int32_t> cbin;
  std::vector<uint32_t> rows;      // every row (T << w | W) of the layout, block by block in the layout's block order
  int order = 0;                   // sc3_code_order
  std::vector<uint32_t> tseq;      // the non-empty T blocks in the order they lie in memory
  std::vector<uint32_t> tidx;      // [1 << t] place of a T block in tseq (0xffffffff: empty block)
  std::vector<size_t> rowstart;    // [tseq.size() + 1] first row of a block inside `rows`
  std::vector<uint64_t> w_nb;
  void *d_w_nb = nullptr;
  std::vector<uint16_t> lo_rlo, lo_rhi;
  void *d_lo_rlo = nullptr, *d_lo_rhi = nullptr;
  void *d_ibase = nullptr, *d_nbase = nullptr, *d_icoff = nullptr, *d_ncoff = nullptr, *d_lo_pat = nullptr,
       *d_w_pat = nullptr, *d_lo_rank = nullptr, *d_w_rank = nullptr, *d_cbin = nullptr, *d_rows = nullptr, *d_nck = nullptr;
  Sc3Layout() = default;
  Sc3Layout(const Sc3Layout &) = delete;
  Sc3Layout &operator=(const Sc3Layout &) = delete;
  ~Sc3Layout();
  // 0 on success; want_device: upload the tables
  int init(int L, int k, int a, int w, bool want_device, int order = 0);
  // is T one of the blocks [b0, b1) of the block sequence?
  bool in_range(uint32_t T, uint32_t b0, uint32_t b1) const {
    return T < tidx.size() && tidx[T] != 0xffffffffu && tidx[T] >= b0 && tidx[T] < b1;
  }
};

// does a (L, k, a, w) combination describe a usable layout?
bool sc3_valid(int L, int k, int a, int w);
// shared, cached layouts (a process uses a handful): the pointer stays valid for the life of the process
const Sc3Layout *sc3_get(int L, int k, int a, int w, bool want_device, int order = 0);

// ---- vector-level operations (sc3_kernels.hip) --------------------------------------------------------------
// All of them act on the part of a vector that covers the blocks [T0, T1) OF THE LAYOUT'S BLOCK SEQUENCE (Sc3Layout::tseq;
// places in it, not values of T -- default: everything): the vectors start at that range's first position of the internal
// layout / first index of the reference order (block order 0; in any other order only whole vectors have a reference side).
// dst (internal) <- src (reference order) when to_internal, else dst (reference order) <- src (internal);
// padding of an internal destination is zeroed
int sc3_layout_copy(const Sc3Layout &Ly, void *dst, const void *src, bool to_internal, hipStream_t st, uint32_t T0 = 0,
                    uint32_t T1 = 0xffffffffu,
    const Sc3Perm *perm = nullptr);
// same for a real array (the cached diagonal)
int sc3_layout_copy_f64(const Sc3Layout &Ly, double *dst, const double *src, bool to_internal, hipStream_t st,
                        uint32_t T0 = 0, uint32_t T1 = 0xffffffffu,
    const Sc3Perm *perm = nullptr);
int sc3_zero_padding(const Sc3Layout &Ly, void *x, hipStream_t st, uint32_t T0 = 0, uint32_t T1 = 0xffffffffu);
// pos[i] = local internal position of the local reference index idx[i] (device arrays)
int sc3_positions(const Sc3Layout &Ly, int64_t n, const int64_t *idx, int64_t *pos, hipStream_t st, uint32_t T0 = 0,
                  uint32_t T1 = 0xffffffffu,
    const Sc3Perm *perm = nullptr);
// counter-based normal deviates keyed by the global reference index (the numbers reference order would get), padding zero
int sc3_random(const Sc3Layout &Ly, void *x, uint64_t seed, hipStream_t st, uint32_t T0 = 0, uint32_t T1 = 0xffffffffu,
    const Sc3Perm *perm = nullptr);
// real vectors of the layout (one double per position): normal deviates / the complex128 vector a real one stands for
int sc3_random_real(const Sc3Layout &Ly, double *x, uint64_t seed, hipStream_t st, uint32_t T0 = 0, uint32_t T1 = 0xffffffffu,
    const Sc3Perm *perm = nullptr);
int sc3_unpack_real(const Sc3Layout &Ly, void *dst, const double *src, hipStream_t st, uint32_t T0 = 0, uint32_t T1 = 0xffffffffu);

// ---- the operator in this layout ---------------------------------------------------------------------------------
// One off-diagonal term of a pair-hop operator on any bond graph (sc3g_kernels.hip), in the layout's bit labelling: the
// XOR mask split by field and the two matrix elements.  The term acts on a ket iff popcount(ket & mask) == half (the
// result keeps k ones); `up` applies when the direction bit -- the lower spin of the pair -- is set in the ket (the
// down spin moves up), `dn` otherwise: ScMask's convention with (lo, lo + 1) replaced by any pair.
struct Sc3Hop {
  uint32_t mT, mW, mLo;
  int32_t half;
  int32_t dfield, dbit;            // field of the direction bit (0 Lo, 1 W, 2 T; 3: `up` always) and its place inside it
  int32_t pad0, pad1;
  double up_re, up_im, dn_re, dn_im;
};

// Per-operator device data of the tiled passes.  A chain bond b couples spins b, b+1 with the two matrix elements of
// ScMask (kernels.h): `up` when the ket has bit b set and bit b+1 clear, `dn` for the opposite hop.
struct Sc3Op {
  const double *bond = nullptr;    // [(L-1) * 4] up_re, up_im, dn_re, dn_im
  uint64_t present = 0;            // bonds the operator has
  uint64_t bondsA = 0, bondsB = 0; // bonds outside the LDS tile gathered by the lo pass / the window pass
  const double *diag = nullptr;    // cached diagonal, internal order (DIAGM 1)
  const double *dlo = nullptr;     // DIAGM 2: the diagonal terms that see Lo only, by (kl, lr) like lo_pat
  const uint64_t *dt_sign = nullptr;   // DIAGM 2: the other terms: sign mask >> a, coefficient, group (0: the term
  const double *dt_coef = nullptr;     //   sees (T, W) only; j >= 1: it also sees Lo through glo[j-1])
  const int32_t *dt_group = nullptr;
  int32_t ndt = 0, ngroups = 0;
  uint32_t glo[4] = {0, 0, 0, 0};
  // operators on any bond graph: the off-diagonal terms by the pass and the way they are applied -- from the LDS tile
  // (both spins inside Lo / inside W) or gathered (lo pass: one spin in Lo, the other in W or T; window pass: none in Lo)
  const Sc3Hop *ldsA = nullptr, *gatA = nullptr, *ldsB = nullptr, *gatB = nullptr;
  int32_t nldsA = 0, ngatA = 0, nldsB = 0, ngatB = 0;
  // window pass, its LDS hops: wnb[(w_off[cw] + wr) * nldsB + h] = rank of the row hop h couples row wr of class cw to,
  // nw[cw] (the zero row behind the tile) where it does not act
  const uint8_t *wnb = nullptr;
  // lo pass, its LDS hops by table (at most SC3G_MAX_PTAB of them): ptab[(ptab_row[kl] + r) * nhp + h] = byte offset in
  // the row's LDS tile (entries of 16 bytes; 8 in real arithmetic) of the entry hop h couples entry r of class kl to --
  // the entry behind the row's nl[kl] entries, which the kernels keep at zero, where the hop does not act.  nhp = the
  // hops rounded up to 8 (a table row is whole 16-byte words); a class has nl[kl] + 1 rows rounded up to even.  One
  // 16-bit extract per (entry, hop) in place of a pair test and a two-table rank: the pass was bound by those
  // instructions (profiles/r05_kagome_real_counters.txt).  Null: the rank tables.
  const uint16_t *ptab = nullptr;
  const double *pcoef = nullptr;   // [nhp] the hops' matrix elements (up_re; zeros behind the last hop): direction-independent real operators
  int32_t nhp = 0;
  int32_t ptab_row[SC3_MAXA + 2] = {};
};
// Per-call data: partition offsets (x holds the internal positions [win_start, ...), y / diag / z are this rank's
// vectors starting at internal position row0), start vectors and fused sums as in launch_sc_block
struct Sc3Call {
  int64_t row0 = 0, win_start = 0;
  const double2 *zinit = nullptr;
  double zscale = 0.0;
  const double2 *zinit2 = nullptr;
  double z2re = 0.0, z2im = 0.0;
  double *dot_out = nullptr;       // 3 * sc3_dot_partials() doubles: per-workgroup <x,y> (re, im) and |y|^2
};

// Partition of the layout over ranks: rank r owns the blocks [Tb[r], Tb[r+1]) of the block sequence (whole blocks, balanced
// by internal length), i.e. a contiguous range of the internal layout -- and, in block order 0, of the reference order.
// Tb has nranks + 1 entries.
std::vector<uint32_t> sc3_partition(const Sc3Layout &ly, int nranks);
// internal / reference offsets of the block range [T0, T1) of the sequence: start and length (block orders other than 0:
// *nstart = -1 unless the range starts the sequence; *nlen = the states in the range)
void sc3_range(const Sc3Layout &ly, uint32_t T0, uint32_t T1, int64_t *istart, int64_t *ilen, int64_t *nstart, int64_t *nlen);

struct Sc3Mat {
  const Sc3Layout *ly = nullptr;
  uint32_t T0 = 0, T1 = 0;         // the blocks of the layout's sequence this rank's rows cover
  int64_t row0 = 0;                // internal position of its first row
  std::vector<uint32_t> rowsel;    // its rows (T << w | W), for the row kernel
  void *d_rowsel = nullptr;
  std::vector<char> needT;         // T blocks its rows read (own blocks included)
  bool tiled = false;              // two tiled passes (every off-diagonal mask is a pair hop); else the row kernel
  bool graph = false;              // ... of sc3g_kernels.hip (any bond graph); false: the chain kernels
  bool sym = false;                // every bond real and direction-independent
  Sc3Perm perm;                    // site relabelling of the vectors (the operator arrays handed to init are in it already)
  bool real = false;               // real vectors (DNM_MAT_REAL_PACKED): sc3_lo_pass_r, window pass on the halved tables
  int diag_mode = 0;               // 0: no diagonal terms; 2: on the fly; 1: needs the cached diagonal
  Sc3Op op{};
  std::vector<uint32_t> permA, permB;
  void *d_permA = nullptr, *d_permB = nullptr, *d_bond = nullptr, *d_dlo = nullptr, *d_dt_sign = nullptr,
       *d_dt_coef = nullptr, *d_dt_group = nullptr, *d_hops = nullptr, *d_wnb = nullptr, *d_ptab = nullptr, *d_pcoef = nullptr;
  std::vector<Sc3Hop> hops;        // ldsA, gatA, ldsB, gatB back to back
  std::vector<uint8_t> wnb;
  std::vector<uint16_t> ptab;
  std::vector<double> pcoef;
  Sc3Mat() = default;
  Sc3Mat(const Sc3Mat &) = delete;
  Sc3Mat &operator=(const Sc3Mat &) = delete;
  ~Sc3Mat();
  int init(const Sc3Layout *layout, const std::vector<int64_t> &masks, const std::vector<int64_t> &mask_offsets,
           const std::vector<int64_t> &signs, const std::vector<double> &rcoef, const std::vector<ScMask> &scm,
           bool want_device, uint32_t T0, uint32_t T1, bool real_vectors = false);
  // the internal positions [lo, hi] this rank's rows read; marks the chunks of 2^shift positions among them
  void window(int64_t *lo, int64_t *hi) const;
  void chunks(int shift, int64_t first_chunk, int64_t nchunks, uint8_t *map) const;
  // ... and exactly: the needed blocks merged into maximal runs of positions [lo, hi), ascending
  std::vector<std::pair<int64_t, int64_t>> ranges() const;
};
bool sc3_instance(int a, int w);           // kernel instances exist for this field split
// the two passes for operators on any bond graph (sc3g_kernels.hip); same contract as launch_sc3's tiled branch
constexpr int SC3G_MAX_GATHER = 64, SC3G_MAX_WLDS = 40;
constexpr int SC3G_MAX_PTAB = 32;     // LDS hops of the lo pass the partner table serves (four 16-byte words per row)
int launch_sc3g(const Sc3Mat &M, const Sc3Call &call, const double *cached_diag, const void *xw, void *y, hipStream_t st,
                int phase);
size_t sc3_dot_partials(const Sc3Mat &M);
// y = A x (- zscale zinit + z2 zinit2), fused sums if asked for; cached_diag: internal order or null.
// phase 0: everything; tiled operators also split: phase 1 = the part whose columns a rank owns itself (the lo pass:
// bonds inside Lo, the Lo/W boundary, the diagonal, the start vectors; writes y), phase 2 = the rest (the window
// pass, adds to y) -- a partitioned multiply runs phase 1 while the window of x is assembled
int launch_sc3(const Sc3Mat &M, const DevMsc &msc, const Sc3Call &call, const double *cached_diag, const void *xw,
               void *y, hipStream_t st, int phase = 0);

}  // namespace dnm
