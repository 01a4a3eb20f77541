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
};

// Operator in row-evaluated index-space form.
struct OpForm {
  int n = 0;                     // index-space bits
  std::vector<RowMask> masks;    // sorted by mask; masks[0].mask == 0 is the diagonal
};

// ---- device tables (plain structs, read with scalar loads) -----------------
struct DevTerm {
  uint64_t sign_ext;   // sign bits outside the tile (global positions, incl. rank bits)
  uint32_t sign_tile;  // sign bits inside the tile, in tile coordinates
  uint32_t pad;
  double coeff;
};

enum : uint32_t { MF_GATHER = 1, MF_KVAR = 2 };

struct DevMask {
  uint32_t mask_tile;   // tile masks: flipped bits in tile coordinates
  uint32_t mask_loc;    // gather masks: flipped bits of the local index (global positions)
  uint32_t re_begin, re_end;   // real terms   [re_begin, re_end)
  uint32_t im_begin, im_end;   // imag terms   [im_begin, im_end)
  uint32_t flags;
  uint32_t src;         // gather source slot
};

struct DevPass {
  // geometry: tile coordinate bits [seg_off[j], seg_off[j]+seg_len[j]) <-> local
  // index bits starting at seg_pos[j]; block-id bits likewise for the rest.
  int32_t nseg;
  int32_t seg_off[MAXSEG], seg_len[MAXSEG], seg_pos[MAXSEG];
  int32_t nbseg;
  int32_t bseg_off[MAXSEG], bseg_len[MAXSEG], bseg_pos[MAXSEG];
  uint64_t sign_base;   // constant OR-ed into the row for sign evaluation (rank bits)
  int32_t accumulate;   // 0: y = ..., 1: y += ...
  int32_t need_tile;    // 0: no tile mask and no diagonal -> skip the LDS stage
  int32_t has_diag;
  // diagonal terms: tile-external ones, then per-k-bucket lists
  uint32_t dext_begin, dext_end;
  uint32_t dbucket[MAXR + 1];
  int32_t nmasks;
  const DevMask *masks;
  const DevTerm *terms;
};

// ---- host-side description --------------------------------------------------
struct PassSpec {
  int B = 0;                       // tile bits
  int nseg = 0;
  int seg_len[MAXSEG] = {0}, seg_pos[MAXSEG] = {0};
  std::vector<int> tile_masks;     // indices into OpForm.masks served from LDS
  std::vector<int> gather_masks;   // indices served by global gathers
  std::vector<int> gather_src;     // source slot per gather mask
  bool has_diag = false;
  bool accumulate = false;
  int partner = -1;                // remote pass: partner rank, else -1
  uint64_t tile_bits() const {
    uint64_t m = 0;
    for (int j = 0; j < nseg; ++j) m |= (((uint64_t)1 << seg_len[j]) - 1) << seg_pos[j];
    return m;
  }
};

struct PlanConfig {
  int B = 12;          // log2 tile amplitudes
  int logR = 4;        // log2 rows per thread
  int amin = 3;        // smallest allowed low segment (2^amin * 16 B contiguous runs)
  int mode = 0;        // 0: multi-pass LDS tiles; 1: single pass, everything else gathered
  int max_gather_span = 0;   // mode 0: masks the tiler cannot place are gathered
};

struct Plan {
  int n = 0, n_loc = 0;
  int rank = 0, nranks = 1;
  PlanConfig cfg;
  std::vector<PassSpec> local;                 // passes on x_local
  std::vector<PassSpec> remote;                // one per partner rank
  std::vector<int> partners;                   // partner rank per remote pass
  bool use_tiled = false;                      // false: generic row-gather kernel only
  std::string describe(const OpForm &op) const;
};

PlanConfig plan_config_from_env();
int make_plan(const OpForm &op, int rank, int nranks, const PlanConfig &cfg, Plan *out);

}  // namespace dnm
