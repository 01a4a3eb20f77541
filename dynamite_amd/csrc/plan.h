// Execution plan of one matrix-free multiply y = H x on a hypercube index
// space (Full, or Parity mapped onto L-1 bits), and the device-side tables the
// tiled kernel reads.
//
// Vocabulary
//   index space : n bits; index i <-> amplitude x[i].  For Full n = L and the
//                 index IS the spin configuration; for Parity n = L-1 and the
//                 dropped bit is folded into the sign masks (see opform.cpp).
//   n_loc       : bits of the index that address the local vector (n minus
//                 log2(nranks) when the state is partitioned).
//   tile        : 2^B amplitudes staged in LDS by one workgroup; its bits are a
//                 union of up to MAXSEG contiguous bit ranges of the local
//                 index ("segments"), segment 0 starting at bit 0.
//   pass        : one kernel launch over the whole local vector with one tile
//                 shape; handles the masks whose flipped bits all lie inside
//                 the tile from LDS ("tile masks") and optionally further
//                 masks by coalesced global gathers ("gather masks").
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "dnm_common.h"

namespace dnm {

constexpr int MAXSEG = 4;
constexpr int MAXBSEG = 8;    // block-id bit ranges
constexpr int MAXSRC = 8;     // gather sources: 0 = x, 1.. = received partner vectors
constexpr int MAXR = 16;

// One Pauli-string term with the sign evaluated on the ROW index:
// contributes  coeff * (-1)^popcount(row & sign)  to the (real or imaginary
// part of the) matrix element H[row, row ^ mask].
struct RowTerm {
  uint64_t sign;
  double coeff;
  int is_imag;
};

struct RowMask {
  uint64_t mask;                 // index-space bits flipped (global, incl. rank bits)
  std::vector<RowTerm> terms;
  // mask == 0 normally means "the diagonal" (real terms, evaluated by the
  // Walsh-Hadamard path).  Mixed Parity pairs can map an imaginary term onto
  // index mask 0; those live in a second mask-0 entry treated like any other mask.
  bool zero_mask_offdiag = false;
  // Real-packed operators (DNM_MAT_REAL_PACKED, mat.cpp pack_opform): the index space has lost its bit 0, which now
  // distinguishes the real and the imaginary part of an element.  is_imag of a term then names the LANE its
  // coefficient belongs to (0: rows with bit 0 clear = real parts, 1: imaginary parts) and pack_flip says whether the
  // mask flipped bit 0, i.e. whether a lane reads the partner element's OTHER lane.
  bool pack_flip = false;
};

// Operator in row-evaluated index-space form.
struct OpForm {
  bool packed = false;           // real-packed form (see RowMask::pack_flip)
  int n = 0;                     // index-space bits
  std::vector<RowMask> masks;    // sorted by mask; masks[0].mask == 0 is the diagonal
};

// ---- device tables (plain structs, wave-uniform, read with scalar loads) ------
// One record carries up to four terms that share a mask: slots 0,1 add to the
// real part of the matrix element, slots 2,3 to the imaginary part (diagonal
// records: all four slots are real terms).  Unused slots have coeff == 0.
struct DevQuad {
  uint32_t mask_tile;     // tile masks: flipped bits in tile coordinates
  uint32_t mask_loc;      // gather masks: flipped bits of the local index (global positions)
  uint32_t src;           // gather source slot (0 = x, 1 = partner vector)
  uint32_t nslots;        // diagonal records: populated slots (1..4)
  uint32_t sign_tile[4];  // sign bits inside the tile, in tile coordinates
  uint64_t sign_ext[4];   // sign bits outside the tile (global positions, incl. rank bits)
  double coeff[4];
};

// A mask with MANY terms (SYK: 16 Majorana products per set of four flipped spins) whose sign masks agree outside the
// flipped bits: s_t = z ^ sigma_t, sigma_t inside the mask.  Its matrix element is then
//     c(row) = (-1)^popcount(row & z) * F[the row's bits at the flipped positions],
// F a table of 2^nbits complex numbers computed once on the host (the Walsh-Hadamard sum of the terms over the flipped
// bits) -- one parity, one table look-up and one complex multiply per row instead of a sign, an add and a multiply per
// TERM.  Masks whose terms fall into several such groups get one record per group.
constexpr int MAXTABBITS = 4;
// Everything the kernel would have to DECIDE per record is decided here, so that its loop is straight-line code (the first
// form branched per table bit on where the bit lives: 1.4 scalar / branch instructions per vector instruction): a row of a
// thread is (block part | tid | k << LOGNT); table bit q comes from exactly one of the three --
//   thread part: bit-field extract of tid at tpos[q], width twid[q] (0: not a thread bit),
//   block part : bit epos[q] of the global row index, width ewid[q],
//   k part     : precomputed per row of a thread, nibble k of `ik`.
struct DevTab {
  uint32_t mask_tile;     // tile masks: flipped bits in tile coordinates
  uint32_t mask_loc;      // gather masks: flipped bits of the local index (global positions)
  uint32_t src;           // gather source slot
  uint32_t first;         // first entry of the table in DevPass::tabvals (2^(flipped bits) entries)
  uint32_t z_tile;        // common sign bits on the THREAD part of the tile coordinate
  uint32_t flags;         // bit 0: some table bit is a k bit (one look-up per row instead of per thread);
                          // bit 1: last record of its mask (the groups of a mask are consecutive and share the partner fetch)
  uint32_t tpos, twid;    // byte q: see above
  uint32_t epos, ewid;
  uint64_t z_ext;         // common sign bits outside the tile (global positions, incl. rank bits)
  uint64_t ik;            // nibble k: table-index bits of row k of a thread
  uint32_t ksign;         // bit k: parity of the common sign mask over the k bits of row k
  uint32_t nbits;         // flipped bits = bits of the table index (<= MAXTABBITS), in ascending index position
};

// off-diagonal record ranges, in table order
enum {
  LP_TILE_REAL_K0 = 0,   // LDS, real coefficient, k-invariant, partner keeps this thread's k (immediate LDS offsets)
  LP_TILE_REAL,          // LDS, real, k-invariant
  LP_TILE_CPLX,          // LDS, complex, k-invariant
  LP_TILE_KVAR_REAL,     // LDS, real, sign reaches the k bits
  LP_TILE_KVAR_CPLX,
  LP_GATHER_REAL,        // global gather, real, k-invariant
  LP_GATHER_KVAR_REAL,   // global gather, real, sign reaches the k bits
  LP_GATHER_CPLX,        // global gather with an imaginary part, k-invariant
  LP_GATHER_KVAR_CPLX,
  LP_COUNT
};

struct DevPass {
  // geometry: tile coordinate bits [seg_off[j], seg_off[j]+seg_len[j]) <-> local
  // index bits starting at seg_pos[j]; block-id bits likewise for the rest.
  int32_t nseg;
  int32_t seg_off[MAXSEG], seg_len[MAXSEG], seg_pos[MAXSEG];
  int32_t nbseg;
  int32_t bseg_off[MAXBSEG], bseg_len[MAXBSEG], bseg_pos[MAXBSEG];
  uint64_t sign_base;   // constant OR-ed into the row for sign evaluation (rank bits)
  int32_t accumulate;   // 0: y = ..., 1: y += ...
  int32_t need_tile;    // 0: no tile mask and no diagonal -> skip the LDS stage
  int32_t has_diag;
  int32_t cache_policy; // bit0: write y through L2 (sc1 stores, line not kept); bit1: non-temporal y loads;
                        // bit2: non-temporal x tile loads; bit6: non-temporal y stores;
                        // bit5: gathers before the barrier, right behind
                        // the tile loads (default); bit7: an accumulating pass loads its y right before the
                        // stores and adds it, instead of starting from it (default: y then crosses the L2 after
                        // the workgroup's gathers -- L=30: 16.90 -> 16.74 ms, profiles/r02_exp44_late_y.txt)
  // diagonal records: tile-external terms, then one list per k-bucket
  uint32_t dext_begin, dext_end;
  uint32_t dbucket[MAXR + 1];
  uint32_t loop[LP_COUNT + 1];   // record range of loop i: [loop[i], loop[i+1])
  int32_t nquads;
  int32_t n_eff;        // index bits this pass runs over (n_loc, or n_loc - 1 for a half-block partner pass)
  const DevQuad *quads;
  double *dot_out;      // non-null (last pass, tile staged): per-workgroup partial sums of conj(x_row) y_row (re, im) and |y_row|^2
  const void *zinit;    // non-null (first, non-accumulating pass): y starts from -zscale * zinit (Lanczos: the
  double zscale;        //   beta term of the three-term recurrence rides on the multiply)
  const void *zinit2;   // with zinit: a second start vector with a complex factor, y += (z2re + i z2im) * zinit2
  double z2re, z2im;    //   (Clenshaw's a_k x term)
  int32_t tile_bits;    // B and LOGR of the kernel instance this pass runs on (passes of one plan may differ)
  int32_t log_rows;
  // XOR-swizzled vector layout: element i of a vector of this space lives at i ^ (((i >> swz_shift) &
  // (2^(swz_shift-4) - 1)) << 4); 0 = natural order.  Sub-block (partner) passes address y relative to the
  // sub-block and read a slice of the partner's block: the swizzle of the fixed offset bits is a constant XOR.
  int32_t swz_shift;
  uint32_t swz_xor_y, swz_xor_src;
  uint32_t block_offset;   // first workgroup of a launch over a RANGE of the pass's workgroups (dnm_mat_mult_local_part)
  // Addressing: the position of a row is pos(block part) ^ pos(thread's tile coordinate) ^ pos(k bits); pos_tmask holds
  // every position bit the THREAD part can reach (the low B - LOGR tile bits and what the layout folds them onto), all
  // below bit 28: the kernel adds everything outside the mask to the scalar base pointer and keeps a 32-bit byte offset
  // per thread (build_pass raises LOGR of a pass whose tile reaches higher until its thread bits fit).
  uint32_t pos_tmask;
  // diagonal terms whose sign mask lies entirely inside the tile do not depend on the block: their sum per tile
  // coordinate, 2^B doubles (32 KB at B = 12, L2-resident), computed once on the host -- one 8-byte load per
  // amplitude instead of ~25 vector instructions.  Terms that see the tile AND bits outside it stay in the
  // k-bucket lists; terms outside the tile in the dext list.  Null: every tile term is in the bucket lists.
  const double *dtile;
  // table records (DevTab): [tab_loop[0], tab_loop[1]) read their partners from the LDS tile, [tab_loop[1], tab_loop[2])
  // gather them; tabvals: their tables, (re, im) pairs.  Passes with any run on the kernel instance that knows them.
  const DevTab *tabs;
  const double *tabvals;
  uint32_t tab_loop[3];
  uint32_t pad_tab;
  // Grouped diagonal terms (same kernel instance as the table records): diagonal terms that see the tile AND bits outside it,
  // grouped by their sign mask INSIDE the tile.  A group's sum over the outside bits is the same for the whole workgroup
  // (an all-to-all ZZ coupling: for every spin of the tile one sum over the 16 spins outside it): computed once per
  // workgroup into LDS, the threads then see one term per GROUP.  Group g = record quads[gbucket[0] + g]: sign_tile[0] =
  // the group's in-tile sign mask, mask_loc = its first term record, src = how many (terms four to a record, sign_tile 0);
  // the groups are listed by the k part of their sign mask: bucket j = [gbucket[j], gbucket[j + 1]).
  uint32_t gbucket[MAXR + 1];
  uint32_t pad_g;
};
constexpr uint32_t MAXDGROUPS = 64;

// ---- host-side description --------------------------------------------------
struct PassSpec {
  int B = 0;                       // tile bits
  int logR = 0;                    // log2 rows per thread (0: the plan's)
  int nseg = 0;
  int seg_len[MAXSEG] = {0}, seg_pos[MAXSEG] = {0};
  std::vector<int> tile_masks;     // indices into OpForm.masks served from LDS
  std::vector<int> gather_masks;   // indices served by global gathers
  std::vector<int> gather_src;     // source slot per gather mask
  bool has_diag = false;
  bool accumulate = false;
  int partner = -1;                // remote pass: partner rank, else -1
  // Remote passes run over a sub-block of the rank's rows: 2^n_eff rows starting at
  // row y_off, reading the 2^n_eff amplitudes that start at src_off of the partner's
  // block.  sign_extra holds the index bits fixed inside the sub-block.
  int n_eff = 0;                   // 0: the whole local block (n_loc bits)
  int64_t y_off = 0, src_off = 0;
  uint64_t sign_extra = 0;
  // XCD group: local index bits [gpos, gpos+glen) are mapped to the block-id bits
  // just above the XCD selector, so the workgroups resident on one XCD at a time
  // span them and gathers across these bits are served by that XCD's L2.
  int glen = 0, gpos = 0;
  uint64_t tile_bits() const {
    uint64_t m = 0;
    for (int j = 0; j < nseg; ++j) m |= (((uint64_t)1 << seg_len[j]) - 1) << seg_pos[j];
    return m;
  }
};

struct PlanConfig {
  int B = 12;          // log2 tile amplitudes
  int logR = -1;       // log2 rows per thread (-1: 4 when the local vector has >= 2^26 amplitudes, else 3)
  int amin = -1;       // smallest allowed low segment (2^amin * 16 B contiguous runs); -1: see make_plan
  int mode = 2;        // 0: multi-pass LDS tiles; 1: single pass, everything else gathered;
                       // 2: multi-pass LDS tiles + L2-served gathers over an XCD group
  int gbits = -1;      // mode 2: bits per XCD group (-1: 9 / 8 / 6 for local vectors of >= 2^30 / >= 2^26 / fewer amplitudes)
  int Bw = 0, logRw = 0; // mode 2 experiment: tile bits / rows per thread of the window passes (0: as B / logR)
  int window_first = -1; // run the window passes before the contiguous one, which then accumulates (-1: with swizzled
                         // vectors of >= 2^25 local amplitudes; DNM_WINDOW_FIRST)
  int gbits_window = -1; // mode 2: cap of the group bits of window-tile passes (-1: no cap)
  int cache_policy = 226; // DevPass::cache_policy for every pass; default: gathers right behind the tile loads (32) + streaming loads (2) and stores (64) of y + an accumulating pass adds its y at the end (128)
  int max_gather_span = 0;   // mode 0: masks the tiler cannot place are gathered
  int diag_last = -1;        // mode 2: evaluate the diagonal in the last local pass instead of the first (-1: with
                             // swizzled vectors; DNM_DIAG_PASS=first|last)
  int swz = 0;               // XOR-swizzle shift of the vectors this plan multiplies (0: natural order)
};

struct Plan {
  int n = 0, n_loc = 0;
  int rank = 0, nranks = 1;
  PlanConfig cfg;
  std::vector<PassSpec> local;                 // passes on x_local
  std::vector<PassSpec> remote;                // one per received (partner, sub-block)
  // what the partners need from this rank, in the order they post their receives
  struct Send { int partner; int64_t offset, count; };
  std::vector<Send> sends;
  bool use_tiled = false;                      // false: generic row-gather kernel only
  std::string describe(const OpForm &op) const;
};

PlanConfig plan_config_from_env();
int make_plan(const OpForm &op, int rank, int nranks, const PlanConfig &cfg, Plan *out);

}  // namespace dnm
