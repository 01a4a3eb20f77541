// Launch interfaces of the HIP kernels (defined in matvec_kernels.hip and
// vec_kernels.hip).  Everything here is host-callable and asynchronous on the
// given stream.
#pragma once

#include <vector>

#include "plan.h"
#include "subspace.h"

namespace dnm {

// Generic operator tables for the row-gather kernels (any subspace pair):
// the reference's shell_context arrays (shell_context.h:12-27) on the device.
struct DevMsc {
  int32_t nmasks;
  const int64_t *masks;
  const int64_t *mask_offsets;
  const int64_t *signs;
  const double *real_coeffs;   // one double per term (bpetsc_template_2.c:286-290)
};

bool tile_config_supported(int B, int logR);

// One pass of the tiled hypercube kernel over the local vector.
// nparts > 1: only the 1/nparts of the workgroups that starts at P.block_offset
int launch_tile_pass(const DevPass &P, int B, int logR, bool glds, int n_loc,
                     const void *x, void *y, const void *xr, hipStream_t st, unsigned nparts = 1);

// y (+)= H x by one thread per row with index maps (MatMult semantics of
// bcuda_template_2.cu:200-273 / bpetsc_template_2.c:371-412).
// Rows [row0, row0 + M); x holds the columns [win_start, ...) in the layout xswz (-1: the right subspace's own);
// colrange != null: no multiply, per-workgroup (min, max) of the columns touched (2 * gather_num_blocks(M) int64).
// column-range sweeps, optional: which chunks of 2^shift columns are read (map[(col >> shift) - first] = 1) -- what a
// partition really has to receive of its column window
struct ColMark {
  uint8_t *map = nullptr;
  int32_t shift = 0;
  int64_t first = 0;
};
int gather_num_blocks(int64_t M);
int gather_rows_per_block();     // workgroup b covers rows [b, b + 1) * gather_rows_per_block()
int launch_gather_matvec(const DevMsc &msc, const SubView &left, const SubView &right,
                         int64_t M, const double *diag, const void *x, void *y,
                         hipStream_t st, int64_t row0 = 0, int64_t win_start = 0, int xswz = -1,
                         int64_t *colrange = nullptr, ColMark mark = ColMark());

// Per-mask precomputation for the SpinConserve kernel.  fast != 0: the mask is a
// bond of two adjacent spins (3 << lo) whose sign masks all lie inside the bond,
// so the matrix element takes one of two values: `up` when the down spin moves
// from site lo to lo+1 (ket has bit lo set), `dn` for the opposite hop.
// pair == 1: the same for any two spins lo < hi (`up`: the ket has bit lo set, the down spin moves to hi); pair == 2
// (XParity on top, subspaces.py:632-674): the hop between spin lo and spin L-1 composed with the global flip -- the
// mask is every spin but those two, it acts on a representative (spin L-1 up) iff spin lo is down, with the element
// `up`; dead != 0: the mask flips an odd number of spins and never keeps a state in the subspace.
struct ScMask {
  int32_t fast;
  int32_t lo;
  double up_re, up_im, dn_re, dn_im;
  int32_t pair, hi, dead, pad;
};
// the table for an operator (host arrays as dnm_mat keeps them)
std::vector<ScMask> sc_masks(const std::vector<int64_t> &masks, const std::vector<int64_t> &mask_offsets,
                             const std::vector<int64_t> &signs, const std::vector<double> &rcoef, int L = 0,
                             bool xparity = false);

// SpinConserve(L,k) on both sides: columns by incremental colex rank (row + delta)
// rows [row0, row0+M); xw holds columns [win_start, ...); y / diag are local.  colrange != null:
// no multiply, only per-workgroup (min, max) of the columns read (2 * sc_num_blocks(M) int64).
// Low half of the unranking: the 16-bit patterns with j ones in ascending (= colex) order start at
// tab[off[j]]; a row's configuration is (16 steps over the high positions) + one lookup here.
struct ScLow {
  const uint16_t *tab;     // 65536 entries
  int32_t off[18];
};
int sc_num_blocks(int64_t M);
int sc_rows_per_block();
int launch_sc_matvec(const DevMsc &msc, const ScMask *scm, const ScLow &low, const SubView &sub, int64_t M, int64_t row0,
                     int64_t win_start, const double *diag, const void *xw, void *y, int64_t *colrange,
                     hipStream_t st, ColMark mark = ColMark());

// Block form of the same product (sc_block_kernel): one workgroup per high part H = state >> lb.
struct ScBlock {
  const uint16_t *lowtab;  // the lb-bit patterns grouped by popcount, ascending inside a group
  int32_t off[18];         // group j starts at lowtab[off[j]]
  int32_t lb;              // low bits per block
  int32_t swizzle;         // XCD-aware block order (needs h_first % 512 == 0)
  int64_t h_first, h_last; // high parts covered by the launch
  const uint32_t *perm;    // optional block order: workgroup b owns high part h_first + perm[b] (0xffffffff: none)
  int64_t nperm;
};
bool sc_block_supported(int lb);
int sc_block_max_masks();
// zinit != null: y = A x - zscale * zinit (local vector); dot_out != null: per-workgroup partial sums of
// conj(x_row) y_row (re, im) and |y_row|^2 -- 3 * sc_block_grid(blk) doubles, zeroed by the launch
int64_t sc_block_grid(const ScBlock &blk);
int launch_sc_block(const DevMsc &msc, const ScMask *scm, const ScBlock &blk, const SubView &sub, int64_t M, int64_t row0,
                    int64_t win_start, int64_t win_len, const double *diag, const void *xw, void *y,
                    hipStream_t st, const void *zinit = nullptr, double zscale = 0.0, double *dot_out = nullptr,
                    const void *zinit2 = nullptr, double z2re = 0.0, double z2im = 0.0);

// diag[row] = sum over mask-0 terms (bcuda_template_1.cu:29-66)
int launch_diag(const DevMsc &msc, const SubView &sub, int64_t M, int64_t row0, double *diag, hipStream_t st);

// per-block maxima of the row sums of |H| (bcuda_template_2.cu:331-403);
// block_max must hold norm_num_blocks(M) doubles.
int norm_num_blocks(int64_t M);
int launch_norm(const DevMsc &msc, const SubView &left, const SubView &right, int64_t M,
                int64_t row0, double *block_max, hipStream_t st);

// CheckConserves: *bad is set to 1 if some column of the right subspace is mapped outside the left one
int launch_conserves(const DevMsc &msc, const double *coeffs_im, const SubView &left, const SubView &right,
                     int64_t N, int *bad, hipStream_t st);

// Reduced density matrix.  Bit layout of a kept / traced configuration: segment i takes the next len[i]
// bits of the compact value and puts them at spin position pos[i] (segments in ascending order).
struct RdmGeom {
  int32_t k, L;               // kept spins, all spins
  int32_t nseg_keep, nseg_tr;
  int8_t klen[33], kpos[33];
  int8_t tlen[33], tpos[33];
};
// workgroup geometry for a given RdmGeom and the scratch it needs
void rdm_plan(const RdmGeom &geo, int *logtm, int *ntiles, int *nsplit, int64_t *chunks_per_split,
              size_t *partial_bytes);
// rho (2^k x 2^k complex128, row-major) from the state x on `sub`; partial: scratch of rdm_plan's size
int launch_rdm(const void *x, const SubView &sub, const RdmGeom &geo, void *partial, void *rho, hipStream_t st);

// ---- vector kernels ---------------------------------------------------------
int vk_set(void *x, int64_t n, double re, double im, hipStream_t st);
int vk_copy(void *y, const void *x, int64_t n, hipStream_t st);      // y = x (complex128 elements; no overlap)
int vk_scale(void *x, int64_t n, double re, double im, hipStream_t st);
int vk_axpby(void *y, const void *x, int64_t n, double are, double aim, double bre, double bim,
             hipStream_t st);
int vk_random(void *x, int64_t n, uint64_t seed, int64_t offset, hipStream_t st, int swz = 0);
int vk_swizzle_copy(void *dst, const void *src, int64_t n, int swz, hipStream_t st);
int vk_unpack_real(void *dst, const void *src, int64_t n_packed, int swz_src, int swz_dst, hipStream_t st);
// partial sums: out_dev[2*nv * nblocks]; reduce_blocks returns the block count
int vk_mdot_blocks(int64_t n);
int vk_sweep_blocks(int64_t n);                     // workgroups of the one-element-per-thread sweeps with partial sums
size_t vk_sweep_scratch(int64_t n, int ncols);     // doubles their partials_dev needs: partials, results, second-level sums
int vk_mdot(const void *V, int64_t ldv, int nv, const void *w, int64_t n, double *partials_dev,
            hipStream_t st);
int vk_maxpy(void *w, const void *V, int64_t ldv, int nv, int64_t n, const double *c_dev,
             hipStream_t st);
int vk_basis_update(void *V, int64_t ldv, int nin, int nout, int64_t n, const double *S_dev,
                    hipStream_t st);
// p = scale * (p - (are + i aim) v - b u) (u may be null); partials_dev[vk_sweep_blocks(n)] then the sum of |p|^2
// (vk_sweep_scratch(n, 1) doubles in all)
int vk_lanczos_update(void *p, const void *v, const void *u, int64_t n, double are, double aim, double b,
                      double scale, double *partials_dev, hipStream_t st);
// y = yscale * y - b z (z may be null) and the sums conj(x) y (re, im), |y|^2: partials_dev[3 * vk_sweep_blocks(n)] then [3]
// (vk_sweep_scratch(n, 3) doubles in all)
int vk_lanczos_dot(void *y, const void *z, const void *x, int64_t n, double b, double *partials_dev,
                   hipStream_t st,
                   double yscale = 1.0);     // y = yscale * y - b z
// out[c] = sum_b partials[b * ncols + c]
size_t vk_reduce_scratch(int ncols);      // doubles of the optional second-level scratch of vk_reduce_partials
int vk_reduce_partials(const double *partials, int nblocks, int ncols, double *out, hipStream_t st, double *tmp = nullptr);
int vk_norm2_partials(const void *x, int64_t n, double *partials_dev, hipStream_t st);

}  // namespace dnm
