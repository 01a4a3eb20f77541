/*
 * dynamite_amd.h -- C ABI of the MI355X-native matrix-free spin-operator engine.
 *
 * This is the drop-in boundary for dynamite's `_backend` shell-matrix path:
 * the entry points below are what dynamite's Cython layer (bpetsc.pyx /
 * bsubspace.pyx) would bind instead of PETSc MatShell + SLEPc MFN/EPS.  Each
 * declaration cites the reference interface it replaces (paths relative to
 * the reference repository root).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure;
 *     dnm_last_error() returns a thread-local message (the reference returns
 *     PetscErrorCode and raises petsc4py.Error, bpetsc.pyx:135-136);
 *   - all integers describing states / masks / indices are int64_t (the
 *     reference's PetscInt under --with-64-bit-indices, bbuild.pyx:28-33);
 *   - state vectors are device pointers to interleaved complex128
 *     (re, im doubles), length = subspace dimension (local part when
 *     partitioned);
 *   - host arrays passed in are BORROWED for the duration of the call; the
 *     handle keeps its own copies (BuildContext deep-copies,
 *     src/dynamite/_backend/bpetsc_template_2.c:275-297);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     Kernels are enqueued asynchronously; functions that return host scalars
 *     synchronise that stream.
 */
#ifndef DYNAMITE_AMD_H
#define DYNAMITE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* errors / device                                                    */
/* ------------------------------------------------------------------ */
const char *dnm_last_error(void);
int dnm_version(void);
int dnm_device_count(int *count);
int dnm_set_device(int device);
/* thin device-memory helpers so a non-torch host (Cython/C) can own vectors;
 * replaces PETSc VecCreate/VecDestroy for VECCUDA
 * (src/dynamite/_backend/bcuda_template_2.cu:110-139 MatCreateVecs_GPU). */
int dnm_malloc(void **dptr, size_t bytes);
int dnm_free(void *dptr);
int dnm_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int dnm_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int dnm_stream_synchronize(void *stream);

/* ------------------------------------------------------------------ */
/* subspaces -- src/dynamite/_backend/bsubspace_impl.h                */
/* ------------------------------------------------------------------ */
/* subspace_type, bsubspace_impl.h:17-23 */
enum { DNM_FULL = 0, DNM_PARITY = 1, DNM_EXPLICIT = 2, DNM_SPIN_CONSERVE = 3 };

/* data_Full / data_Parity / data_SpinConserve / data_Explicit
 * (bsubspace_impl.h:41-44, 95-99, 161-167, 265-272) folded into one POD.
 * Python holders: bsubspace.pyx:68-114 (CFull, CParity, CSpinConserve,
 * CExplicit). */
typedef struct dnm_subspace {
  int32_t type;
  int64_t L;
  int64_t space;               /* Parity: 0 even, 1 odd */
  int64_t k;                   /* SpinConserve */
  int64_t ld_nchoosek;         /* SpinConserve: L+1 */
  const int64_t *nchoosek;     /* SpinConserve: (k+1)*(L+1), [kk*ld+LL] = C(LL,kk) */
  int64_t dim;                 /* Explicit */
  const int64_t *state_map;    /* Explicit: idx -> state */
  const int64_t *rmap_indices; /* Explicit: NULL when state_map is sorted */
  const int64_t *rmap_states;  /* Explicit: sorted states */
  /* Layout of state vectors on this subspace in device memory (no counterpart in the reference: a PETSc Vec is
   * opaque too).  0: element i of a rank's block is at position i.  S >= 5, Full / Parity only: XOR-swizzled,
   * element i is at position  i ^ (((i >> S) & (2^(S-4) - 1)) << 4)  -- index bits [S, 2S-4) folded onto bits
   * [4, S), an involution that keeps 256-byte runs together.  The far-apart runs of the tiled multiply's window
   * passes then spread over the L2 sets.  Every dnm_* call that takes a vector of this
   * subspace expects this layout; dnm_vec_swizzle_copy converts to and from index order.
   * SpinConserve: 0 or a | w << 8 (a low bits, w window bits): the three-field internal layout of
   * dynamite_amd/csrc/sc3.h -- the blocks of equal top bits T = state >> (a + w) stay where the reference order has
   * them, inside a block the rows (T, W) are grouped by popcount(W), ordered by the rank of W, and padded to 128-byte
   * lines (padding holds zeros): a vector then has dnm_vec_layout_size() elements instead of C(L, k).  The two-pass
   * SpinConserve multiply works in this layout; dnm_vec_layout_copy / _positions convert.  Partitions hand whole
   * T blocks to a rank (dnm_vec_layout_partition). */
  int32_t vec_swizzle;
  /* SpinConserve in the internal layout, one rank: NULL, or L entries -- spin i of the reference's labelling is bit
   * site_perm[i] of the states the layout orders (a relabelling of the spins: the subspace is invariant under it).
   * dnm_mat_create rewrites the operator into that labelling, so that the pair hops of an operator on a bond graph
   * (the kagome Heisenberg model of examples/scripts/kagome/run_kagome.py) fall inside the layout's fields wherever
   * the graph allows; vectors pass through it wherever they meet the reference order (dnm_vec_layout_copy,
   * _positions, _set_random).  Everything index-wise at this boundary -- rows of dnm_mat_get_diagonal, indices of
   * dnm_vec_layout_positions, files -- stays in the reference's order.  dnm_sc_choose_site_perm picks one. */
  const int8_t *site_perm;
} dnm_subspace;

/* get_dimension_* (bsubspace.pyx:144-162) -> Dim_* */
int dnm_subspace_dim(const dnm_subspace *s, int64_t *dim);
/* idx_to_state_* (bsubspace.pyx:164-184) -> I2S_*_array; out-of-range idx is an error */
int dnm_idx_to_state(const dnm_subspace *s, int64_t n, const int64_t *idxs, int64_t *states);
/* state_to_idx_* (bsubspace.pyx:186-206) -> S2I_*_array; -1 when not in the subspace */
int dnm_state_to_idx(const dnm_subspace *s, int64_t n, const int64_t *states, int64_t *idxs);

/* Site relabelling (dnm_subspace.site_perm) for an operator with the distinct masks `masks` on SpinConserve(L, .)
 * vectors in the (a, w) internal layout: the assignment of spins to the layout's three fields that leaves the fewest
 * pair hops between fields (host search, deterministic; the identity is kept on ties, so chains stay as they are).
 * fix_top != 0: spin L-1 keeps bit L-1 (what an XParity subspace on top needs).  counts (6 ints, may be NULL): hops
 * of the result inside Lo, inside W, inside T, between Lo-W, Lo-T, W-T.  No counterpart in the reference, whose
 * kernels gather every column (bpetsc_template_2.c:371-412). */
int dnm_sc_choose_site_perm(int L, int a, int w, int64_t nmasks, const int64_t *masks, int fix_top, int8_t *site_perm,
                            int32_t *counts);

/* ------------------------------------------------------------------ */
/* shell matrix -- bpetsc.pyx:78-147, bpetsc_impl.h:42-63             */
/* ------------------------------------------------------------------ */
typedef struct dnm_mat dnm_mat;   /* replaces PETSc Mat(MATSHELL) + shell_context (shell_context.h:12-27) */

/* flags for dnm_mat_create */
enum {
  DNM_MAT_DEFAULT      = 0,
  DNM_MAT_FORCE_GATHER = 1,   /* use only the generic row-gather kernel */
  DNM_MAT_USE_GLDS     = 2,   /* stage LDS tiles with global_load_lds DMA instead of through registers
                                 (measured slower on MI355X for this access pattern; kept for A/B runs) */
  DNM_MAT_AMIN_SHIFT   = 8,   /* flags bits 8..15: log2 of the contiguous run of a window tile (0: the planner's default);
                                 the transposed exchange asks for tiles [0, a) + [f, n) so that sub-pieces are contiguous */
  DNM_MAT_HOST_ONLY    = 4,   /* build the plan and its tables on the host only (no device needed;
                                 diagnostics and CPU tests) -- such a handle cannot multiply */
  DNM_MAT_REAL_PACKED  = 16   /* Real arithmetic for a real-symmetric operator (every matrix element real in the
                                 product basis: Heisenberg, XXZ, Ising, random-field chains ...) on a Full / Parity pair
                                 (one rank or 2^p ranks: the exchange is that of an operator on one index bit less; XParity on top allowed): the handle multiplies REAL vectors of the same dimension, stored two
                                 amplitudes to a complex128 element -- element j holds the amplitudes of indices 2j
                                 (real part) and 2j + 1 (imaginary part), so a vector is dim / 2 elements, 8 bytes per
                                 amplitude, and dnm_mat_sizes reports the halved sizes.  Solver-internal (eigsolve of a
                                 real-symmetric operator needs no complex arithmetic; the reference's PETSc build is
                                 complex throughout).  Also for a SpinConserve pair in the internal layout (chain
                                 operators and operators on bond graphs, XParity on one rank): one double per position, i.e.
                                 dnm_vec_layout_size / 2 complex128 elements (dnm_vec_layout_unpack_real).
                                 dnm_eigsolve on such a handle keeps every inner product real, and
                                 dnm_vec_unpack_real turns a packed vector into the complex128 vector the caller sees.
                                 dnm_mat_create fails (and the caller keeps the complex handle) when the operator has
                                 an imaginary matrix element. */
};

/* Row-block partition of the state vector over `nranks` devices, as PetscSplitOwnership splits it
 * (BuildGPUShell, bcuda_template_2.cu:24-27): dim / nranks rows each, the first dim % nranks ranks one more.
 * Two exchange schemes, chosen by dnm_mat_create (dnm_mat_exchange_plan reports no transfers for the second):
 *   - Full/Full or Parity/Parity on 2^p ranks with blocks of at least one tile: XOR-partner sub-blocks
 *     (dnm_mat_exchange_plan + dnm_mat_mult_local + dnm_mat_mult_remote), MatMult_CPU_Fast's scheme
 *     (bpetsc_template_2.c:787-879);
 *   - every other case -- SpinConserve, Explicit / Auto, projections between different subspaces, Full / Parity
 *     on any other rank count: rows in index order, each rank reads its columns through a window
 *     (dnm_mat_column_window + dnm_mat_mult_window), in place of the scatter-add of MatMult_CPU_General's MPI
 *     branch (bpetsc_template_2.c:413-504). */
typedef struct dnm_partition {
  int32_t rank;
  int32_t nranks;
} dnm_partition;

/* BuildMat(msc, subspaces, GPU_SHELL, xparity, &A)  (bpetsc_impl.h:42,
 * bpetsc.pyx:78-138): masks[nmasks] sorted unique, mask_offsets[nmasks+1],
 * signs[nterms], coeffs[nterms] complex128 interleaved.  part may be NULL
 * (single device).  The operator must be Hermitian in the reference's sense
 * (msc_tools.py:94-118): each term purely real or purely imaginary as
 * TERM_REAL(mask, sign) dictates (bpetsc_impl.h:34). */
int dnm_mat_create(int64_t nmasks, const int64_t *masks, const int64_t *mask_offsets,
                   const int64_t *signs, const double *coeffs,
                   const dnm_subspace *left, const dnm_subspace *right,
                   int xparity, int flags, const dnm_partition *part,
                   dnm_mat **out);
/* CheckConserves(msc, subspaces, xparity, &result)  (bpetsc.pyx:150-193,
 * bpetsc_template_2.c:990-1056): *result = 1 iff every column of the right
 * subspace is mapped into the left subspace (or onto a zero matrix element). */
int dnm_check_conserves(int64_t nmasks, const int64_t *masks, const int64_t *mask_offsets,
                        const int64_t *signs, const double *coeffs,
                        const dnm_subspace *left, const dnm_subspace *right,
                        int xparity, int *result, void *stream);
/* ReducedDensityMatrix(vec, sub_type, sub_data, keep_size, keep, triang, rtn_dim, rtn)
 * (bpetsc_impl.h:52-63, bpetsc_template_1.c:87-165, bpetsc.pyx:245-276): the density
 * matrix of the spins keep[0] < keep[1] < ... of the state x (device pointer, the
 * whole vector), all other spins traced out.  rho: device buffer of 4^keep_size
 * complex128, row-major, bit i of a row/column index = spin keep[i]. */
int dnm_reduced_density_matrix(const void *x, const dnm_subspace *sub, int keep_size,
                               const int64_t *keep, void *rho, void *stream);
/* MATOP_DESTROY -> MatDestroyCtx_GPU (bcuda_template_2.cu:110-139) */
int dnm_mat_destroy(dnm_mat *A);
/* MatGetSize / MatGetLocalSize */
int dnm_mat_sizes(const dnm_mat *A, int64_t *M, int64_t *N, int64_t *m_local, int64_t *n_local);
/* PrecomputeDiagonal(A) (bpetsc.pyx:141-147, bcuda_template_1.cu:4-66).  The
 * tiled Full/Parity kernels evaluate the diagonal on the fly and ignore the
 * cache; the generic kernel uses it exactly as the reference does. */
int dnm_mat_precompute_diagonal(dnm_mat *A, void *stream);
/* copies the cached diagonal (double[m_local]) to the host; error if absent */
int dnm_mat_get_diagonal(dnm_mat *A, double *diag_host, void *stream);
/* MATOP_MULT -> MatMult_GPU (bcuda_template_2.cu:141-198): y = A x, y overwritten.
 * Single device (or a partition whose operator has no off-rank masks). */
int dnm_mat_mult(dnm_mat *A, const void *x, void *y, void *stream);
/* MATOP_NORM, NORM_INFINITY only (bcuda_template_2.cu:275-329); cached in the
 * handle like ctx->nrm.  With a partition this is the LOCAL max; the caller
 * max-reduces over ranks (MPIU_Allreduce(MAX), bpetsc_template_2.c:975) and
 * stores it back with dnm_mat_set_norm. */
int dnm_mat_norm_inf(dnm_mat *A, double *nrm, void *stream);
/* The vector layouts the matrix works in (dnm_subspace.vec_swizzle codes): what the descriptors asked for when the
 * matrix supports it, 0 (reference order) otherwise -- e.g. a SpinConserve descriptor with the internal layout paired
 * with another subspace or under XParity.  The caller converts (dnm_vec_layout_copy) when a
 * vector's layout differs.  With an internal layout dnm_mat_sizes reports the local lengths incl. padding. */
int dnm_mat_layouts(const dnm_mat *A, int *left, int *right);
/* 1 if the handle was built with DNM_MAT_REAL_PACKED and multiplies real vectors (sizes then count complex128 elements) */
int dnm_mat_is_real_packed(const dnm_mat *A, int *packed);
int dnm_mat_set_norm(dnm_mat *A, double nrm);
/* human-readable description of the execution plan (passes, tiles) */
int dnm_mat_plan_describe(const dnm_mat *A, char *buf, size_t buflen);
/* number of kernel launches one dnm_mat_mult issues */
int dnm_mat_plan_launches(const dnm_mat *A, int *n);
/* plan introspection (diagnostics, CPU tests of the planner): pass counts, and
 * a copy of one pass's tables (layouts: dynamite_amd/csrc/plan.h DevPass /
 * DevQuad; pointers inside the copied DevPass are meaningless). */
int dnm_mat_plan_counts(const dnm_mat *A, int *n_local_passes, int *n_remote_passes, int *tiled,
                        int *B, int *logR, int *n_loc);
/* the pass's tabulated in-tile diagonal (2^tile_bits doubles; an error when the pass has none) */
int dnm_mat_export_dtile(const dnm_mat *A, int remote, int idx, double *out, int64_t n);
int dnm_mat_export_pass(const dnm_mat *A, int remote, int idx, void *desc_out, size_t desc_bytes,
                        void *quads_out, size_t quad_bytes, int max_quads, int *nquads);
/* the pass's table records (plan.h: DevTab -- masks of many terms, one table look-up per row instead of one sign per term)
 * and their tables ((re, im) pairs); null buffers: the counts alone */
int dnm_mat_export_tabs(const dnm_mat *A, int remote, int idx, void *tabs_out, size_t tab_bytes, int max_tabs, int *ntabs,
                        double *vals_out, int64_t max_vals, int64_t *nvals);

/* --- partitioned multiply: replaces the VecScatterCreateToAll all-gather of
 * bcuda_template_2.cu:161-171 with an XOR-partner exchange. ---------------- */
/* One block transfer of the exchange: `count` amplitudes starting at `offset`
 * of the SENDER's local vector.  sends[]: what this rank must send (pass_id = -1);
 * recvs[]: what it receives, recvs[i] feeding dnm_mat_mult_remote(A, i, ...).
 * A mask that flips rank bits couples this rank to rank ^ (mask >> log2(n_local));
 * blocks on which its matrix elements vanish identically (e.g. the half of the
 * rows a flip-flop term XX+YY annihilates) are neither sent nor swept, and the
 * rank's block is split in halves so that only the needed half travels.  Between
 * one pair of ranks the sends of one side are listed in the order of the
 * receives of the other, so in-order point-to-point matching is sufficient. */
typedef struct {
  int32_t partner;   /* peer rank */
  int32_t pass_id;   /* recvs: index for dnm_mat_mult_remote; sends: -1 */
  int64_t offset;    /* first amplitude, in the sender's local vector */
  int64_t count;     /* amplitudes (complex128) */
} dnm_xfer;
int dnm_mat_exchange_plan(const dnm_mat *A, int *nsend, dnm_xfer *sends /* may be NULL */,
                          int *nrecv, dnm_xfer *recvs /* may be NULL */);
/* y = A x and dot[0] + i dot[1] = sum_i conj(x_i) y_i in the same sweep (the Lanczos
 * alpha = <v, H v>; what SLEPc does with MatMult + VecDot).  Single rank only. */
int dnm_mat_mult_dot(dnm_mat *A, const void *x, void *y, double *dot /* [2], host */, void *stream);
/* One Lanczos step's multiply: y = A x - b z (z may be NULL), dot[0] + i dot[1] = <x, y> and
 * dot[2] = |y|^2 (beta^2 = |y|^2 - |alpha|^2 without another sweep); the beta term starts the
 * accumulators of the first pass, the sums close the last. */
int dnm_mat_mult_lanczos(dnm_mat *A, const void *x, void *y, const void *z, double b,
                         double *dot /* [3], host */, void *stream);
/* y = A x - b z (three-term recurrences without inner products, e.g. Chebyshev).  Single rank only. */
int dnm_mat_mult_sub(dnm_mat *A, const void *x, void *y, const void *z, double b, void *stream);
/* 1 if dnm_mat_mult_sub / _sub2 cost no extra vector sweep for this operator (tiled and SpinConserve block kernels) */
int dnm_mat_fuses_init(const dnm_mat *A);
/* y = A x - b z + (c_re + i c_im) z2 (Clenshaw's recurrence; z2 may be NULL).  Single rank only. */
int dnm_mat_mult_sub2(dnm_mat *A, const void *x, void *y, const void *z, double b, const void *z2,
                      double c_re, double c_im, void *stream);
/* y = (masks that stay on this rank) x_local; y overwritten */
int dnm_mat_mult_local(dnm_mat *A, const void *x_local, void *y, void *stream);
/* y += (masks served by receive `recv_index`) x_recv, where x_recv holds the
 * recvs[recv_index].count amplitudes received from that partner; y is the
 * rank's whole local vector (the pass offsets into it itself) */
int dnm_mat_mult_remote(dnm_mat *A, int32_t recv_index, const void *x_recv,
                        void *y, void *stream);

/* --- partitioned multiply through a column window -------------------------------------------------
 * Basis indices are split as PetscSplitOwnership does (M / P rows each, the first M % P ranks one more; any P).
 * The columns a rank's rows read lie in a window (SpinConserve: col = row +- binomials, a few blocks wide, see
 * DESIGN.md; Explicit / projections: whatever the masks reach, found by one device sweep); the host assembles
 * that window IN INDEX ORDER from the owners' blocks (send/recv) and multiplies.  y_local is the rank's block of
 * the left vector in that subspace's own layout. */
int dnm_mat_ownership(const dnm_mat *A, int64_t *row0, int64_t *m_local);
/* Where dnm_mat_window_split says so (SpinConserve pairs in the internal layout, two tiled passes), the window multiply
 * comes in two parts so that compute overlaps the exchange (bpetsc_template_2.c:866-873 overlaps assembly and compute
 * block by block): _local needs the rank's own vector only (bonds inside a block of equal top bits, the diagonal) and
 * WRITES y; _remote reads the assembled window and ADDS the bonds that reach other blocks.  y = _local, then _remote. */
int dnm_mat_window_split(const dnm_mat *A, int *supported);
/* Tiled Full / Parity operators: the rank-local passes over one of `nparts` (a power of two) equal ranges of their
 * workgroups.  dnm_mat_local_part_bits tells what a range is: *top_free_bit = the highest index bit outside every tile
 * (range `part` covers the amplitudes whose log2(nparts) index bits ending there equal `part`), *gathers = records
 * that read outside the tile (0: a range reads and writes its own amplitudes only).  The transposed exchange runs its
 * layout-B pass sub-piece by sub-piece this way. */
int dnm_mat_mult_local_part(dnm_mat *A, const void *x, void *y, int part, int nparts, void *stream);
int dnm_mat_local_part_bits(const dnm_mat *A, int *top_free_bit, int *gathers);
int dnm_mat_mult_window_local(dnm_mat *A, const void *x_local, void *y_local, void *stream);
int dnm_mat_mult_window_remote(dnm_mat *A, const void *x_window, int64_t win_start, int64_t win_len, void *y_local,
                               void *stream);
/* Every other window partition (SpinConserve in reference order, Explicit, projection pairs, odd rank counts) overlaps
 * by ROWS: dnm_mat_window_local_rows lists up to max_ranges ranges [r0, r1) of the rank's local rows whose columns all
 * lie in [col_lo, col_hi) -- the rank's own block of x, which sits in the window buffer before anything arrives; none
 * shorter than min_blocks workgroups of 256 rows -- and
 * dnm_mat_mult_window_rows multiplies one range of rows; the caller runs the listed ranges under the exchange and the
 * rest after it (bpetsc_template_2.c:866-873). */
int dnm_mat_window_local_rows(dnm_mat *A, int64_t col_lo, int64_t col_hi, int max_ranges, int min_blocks,
                              int64_t *ranges, int *nranges, void *stream);
int dnm_mat_mult_window_rows(dnm_mat *A, const void *x_window, int64_t win_start, int64_t win_len, void *y_local,
                             int64_t r0, int64_t r1, void *stream);
/* SpinConserve pairs in the internal layout (dnm_mat_layouts) partition differently: a rank owns whole blocks of equal
 * top bits T -- a contiguous range of the internal layout (dnm_vec_layout_partition; balanced to within one block)
 * that is also a contiguous range of the reference order -- and ownership, windows and chunks are expressed in
 * positions of the internal layout: ranks exchange ranges of their vectors as they lie. */
/* inclusive column range this rank's rows read; one device sweep, then cached */
int dnm_mat_column_window(dnm_mat *A, int64_t *cmin, int64_t *cmax, void *stream);
/* which chunks of 2^chunk_shift columns inside that window the rank's rows really read:
 * map[(col >> chunk_shift) - (cmin >> chunk_shift)] = 1, nchunks = (cmax >> chunk_shift) - (cmin >> chunk_shift) + 1
 * bytes on the host.  The window is mostly holes for the far hops of a SpinConserve chain (L=24, k=12 on 8
 * ranks: 1.5 blocks needed of a 4.2-block window): only marked chunks have to be received, the multiply never
 * reads the others (the reference scatters exactly the entries it needs, bpetsc_template_2.c:413-504) */
int dnm_mat_column_chunks(dnm_mat *A, int chunk_shift, uint8_t *map, int64_t nchunks, void *stream);
/* The same exactly, where the library knows it without a sweep (SpinConserve pairs in the internal layout: whole blocks of
 * equal top bits): the positions the rank's rows read as maximal runs [lo, hi), ascending -- ranges[2 i], ranges[2 i + 1];
 * *nranges = their number (call with max_ranges = 0 to size the array; 0 for every other partition: use the chunk map).
 * In the block order that keeps the reference's ranges (dnm_subspace.vec_swizzle bits 16-19 = 0) the chunk map is as
 * good; in the order made for partitions (1) the needed blocks interleave with others and only this list is tight. */
int dnm_mat_column_ranges(dnm_mat *A, int64_t max_ranges, int64_t *ranges, int64_t *nranges);
/* y_local = A[own rows, :] x, x_window holding columns [win_start, win_start + win_len) */
int dnm_mat_mult_window(dnm_mat *A, const void *x_window, int64_t win_start, int64_t win_len,
                        void *y_local, void *stream);

/* ------------------------------------------------------------------ */
/* vector kernels (what PETSc Vec / SLEPc BV supply to the Krylov     */
/* loops; states.py:703-797 on the Python side)                       */
/* ------------------------------------------------------------------ */
int dnm_vec_set(void *x, int64_t n, double re, double im, void *stream);          /* VecSet */
int dnm_vec_copy(const void *x, void *y, int64_t n, void *stream);                /* VecCopy */
int dnm_vec_scale(void *x, int64_t n, double re, double im, void *stream);        /* VecScale */
/* y = alpha x + beta y  (VecAXPBY, states.py:779-797) */
int dnm_vec_axpby(void *y, const void *x, int64_t n, double are, double aim,
                  double bre, double bim, void *stream);
/* out[0..1] = sum_i x_i * conj(y_i)  (VecDot(x,y) = y^H x, states.py:703-719) */
int dnm_vec_dot(const void *x, const void *y, int64_t n, double *out, void *stream);
int dnm_vec_norm2(const void *x, int64_t n, double *out, void *stream);           /* VecNorm */
/* Fills x with N(0,1)+iN(0,1) from a counter-based generator keyed by
 * (seed, global index = offset + i): distribution of State.set_random
 * (states.py:292-316), not its MT19937 stream. */
int dnm_vec_set_random(void *x, int64_t n, uint64_t seed, int64_t offset, void *stream);
/* the same stream of numbers for a vector in the swizzled layout (element i gets what index order would give it) */
int dnm_vec_set_random_swz(void *x, int64_t n, uint64_t seed, int64_t offset, int swizzle, void *stream);
/* dst[i] = src[i ^ sw(i)]: swizzled <-> index order (the map is an involution); dst != src */
int dnm_vec_swizzle_copy(void *dst, const void *src, int64_t n, int swizzle, void *stream);
/* The complex128 vector a real-packed one stands for (DNM_MAT_REAL_PACKED; no counterpart in the reference, whose PETSc
 * build is complex throughout -- what its EPS hands to computations.py:273-284 is what dst holds): src has n_packed
 * elements, element j = the real amplitudes of indices 2j and 2j + 1, in the layout swizzle_packed; dst gets
 * 2 * n_packed elements with zero imaginary parts in the layout swizzle_out; dst != src */
int dnm_vec_unpack_real(void *dst, const void *src, int64_t n_packed, int swizzle_packed, int swizzle_out, void *stream);
/* The same for a SpinConserve subspace in the internal layout: src holds one double per position of the rank's part of
 * the layout (what a DNM_MAT_REAL_PACKED operator of that subspace multiplies; part may be NULL on one rank), dst gets
 * the complex128 vector of the layout with zero imaginary parts; dst != src */
int dnm_vec_layout_unpack_real(const dnm_subspace *s, const dnm_partition *part, void *dst, const void *src, void *stream);
/* Vectors of a SpinConserve subspace in the internal layout (dnm_subspace.vec_swizzle = a | w << 8).  No counterpart
 * in the reference: what a petsc4py Vec of that subspace holds, element by element, is reached through these.
 * size: elements of a vector (rows + padding); copy: to_internal != 0: dst (internal) <- src (reference order, C(L,k)
 * elements), else dst (reference order) <- src (internal), dst != src; zero_padding: after an operation that wrote
 * the padding (VecSet, VecShift); positions: pos[i] = where reference index idx[i] lives (device arrays of n int64);
 * set_random: the numbers dnm_vec_set_random gives reference order, padding zero. */
int dnm_vec_layout_size(const dnm_subspace *s, int64_t *n);
/* the part of such a vector rank `rank` of `nranks` owns: [istart, istart + ilen) of the internal layout =
 * [nstart, nstart + nlen) of the reference order */
int dnm_vec_layout_partition(const dnm_subspace *s, int nranks, int rank, int64_t *istart, int64_t *ilen,
                             int64_t *nstart, int64_t *nlen);
/* the layout's T blocks in the order they lie in memory (the unit a partition shares out, and what moves whole between
 * two layouts that differ in block order only -- vec_swizzle bits 16-19): T[j] = the block's value of the T field,
 * ibase[j] = its first position; ibase[*count] = the layout's size.  max < *count (e.g. 0 with null arrays): the count
 * alone.  T: max entries, ibase: max + 1.  Host tables only. */
int dnm_vec_layout_blocks(const dnm_subspace *s, int64_t max, int64_t *T, int64_t *ibase, int64_t *count);
/* `part` (null: one rank) selects the rank whose part of the vector the pointers hold; indices and positions are
 * then local to that part */
int dnm_vec_layout_copy(const dnm_subspace *s, const dnm_partition *part, void *dst, const void *src, int to_internal,
                        void *stream);
int dnm_vec_layout_copy_f64(const dnm_subspace *s, const dnm_partition *part, double *dst, const double *src,
                            int to_internal, void *stream);
int dnm_vec_layout_zero_padding(const dnm_subspace *s, const dnm_partition *part, void *x, void *stream);
int dnm_vec_layout_positions(const dnm_subspace *s, const dnm_partition *part, int64_t n, const int64_t *idx,
                             int64_t *pos, void *stream);
int dnm_vec_layout_positions_host(const dnm_subspace *s, const dnm_partition *part, int64_t n, const int64_t *idx,
                                  int64_t *pos);  /* host arrays */
int dnm_vec_layout_set_random(const dnm_subspace *s, const dnm_partition *part, void *x, uint64_t seed, void *stream);
/* h[j] = V_j^H w for j < nv (BVDotVec); V = nv vectors of length n, stride ldv elements.
 * h_host: 2*nv doubles. */
int dnm_vec_mdot(const void *V, int64_t ldv, int nv, const void *w, int64_t n,
                 double *h_host, void *stream);
/* w += sum_j c[j] V_j (BVMultVec); c_host: 2*nv doubles */
int dnm_vec_maxpy(void *w, const void *V, int64_t ldv, int nv, int64_t n,
                  const double *c_host, void *stream);
/* V[:, 0:nout) = V[:, 0:nin) * S  (in place, S is nin x nout complex column-major,
 * host): the thick-restart basis update (BVMultInPlace). */
int dnm_vec_basis_update(void *V, int64_t ldv, int nin, int nout, int64_t n,
                         const double *S_host, void *stream);

/* ------------------------------------------------------------------ */
/* Krylov solvers -- computations.py:10-126 (SLEPc MFN expokit) and   */
/* computations.py:128-292 (SLEPc EPS Krylov-Schur, HEP)              */
/* ------------------------------------------------------------------ */
/* Distributed hooks: when non-NULL the solver calls them after every local
 * reduction (sum over ranks of `n` doubles, in place / max) and uses `mult`
 * instead of dnm_mat_mult.  A single-device run passes NULL. */
typedef struct dnm_hooks {
  void *ctx;
  int (*mult)(void *ctx, const void *x, void *y);
  int (*allreduce_sum)(void *ctx, double *buf, int n);
  int (*allreduce_max)(void *ctx, double *buf, int n);
} dnm_hooks;

/* converged reasons (SLEPc MFNConvergedReason / EPSConvergedReason as mapped
 * by computations.py:114-122 and :261-275) */
enum {
  DNM_CONVERGED_TOL = 1,
  DNM_CONVERGED_ITS = 2,        /* MFN_CONVERGED_ITS */
  DNM_DIVERGED_ITS = -1,        /* -> MaxIterationsError */
  DNM_DIVERGED_BREAKDOWN = -2,  /* -> ConvergenceError */
  DNM_DIVERGED_SYMMETRY_LOST = -3
};

typedef struct dnm_solver_stats {
  int32_t reason;
  int32_t its;          /* outer steps (MFN) / restarts (EPS) */
  int32_t matvecs;
  int32_t nconv;        /* EPS only */
  double  err_est;      /* MFN: accumulated local error estimate; EPS: largest measured
                         * |H u - theta u| / |theta| of the returned pairs */
} dnm_solver_stats;

/* y = exp(scale * A) x, scale = (scale_re + i scale_im); evolve() passes
 * scale = -i t (computations.py:89-96).  Sidje-Expokit adaptive Krylov
 * (SLEPc MFNEXPOKIT); because A is Hermitian the basis is built by Lanczos
 * with partial re-orthogonalisation (DNM_EXPM_ORTHO=full: against the whole
 * basis every step, as SLEPc's BV).  tol<=0 -> 1e-8; ncv<=0 -> min(30, N);
 * max_its<=0 -> 100 (SLEPc defaults).  work_limit_bytes bounds the basis
 * allocation (0 = no limit): ncv is reduced to fit.  With ncv <= 0, max_its <= 0
 * and scale_re == 0 (a real time) the driver compares, after every accepted step,
 * what the rest of the interval costs at the step size the error control has
 * settled on with what dnm_expm_chebyshev needs for it (known exactly) and hands
 * the rest over when that is clearly cheaper; the expansion is taken from the
 * start when all of it costs less than one outer step of ncv multiplies, when
 * an earlier hand-over on this operator still applies, or when the basis
 * workspace would have to be acquired for a memory-limited basis and ten Lanczos
 * steps from x show that the norm bound is not loose (DNM_EXPM_HYBRID=0: never). */
int dnm_expm_multiply(dnm_mat *A, const void *x, void *y, int64_t n_local,
                      double scale_re, double scale_im, double tol, int ncv,
                      int max_its, size_t work_limit_bytes, const dnm_hooks *hooks,
                      dnm_solver_stats *stats, void *stream);

/* y = exp(-i t A) x for REAL t by the Chebyshev expansion
 *   exp(-i t A) = sum_k (2 - delta_k0) (-i)^k J_k(r t) T_k(A / r),   r = ||A||_inf >= spectral radius,
 * evaluated with the three-term recurrence on the fused multiply y = A x - b z: one multiply and
 * two thirds of a vector sweep per term, four work vectors, no inner products and no basis --
 * an alternative to dnm_expm_multiply when the Krylov basis is what limits the size or the speed
 * (Operator.evolve(algo='chebyshev')).  The series is cut where the remaining Bessel coefficients sum to
 * less than tol/100 (tol <= 0 -> 1e-8); long times are split into steps of r*t <= 64.  Costs about
 * r*t + 6 (r*t)^(1/3) + 10 multiplies; loose norm bounds (r much larger than the spectral radius)
 * cost proportionally more.  stats->its = number of steps, stats->matvecs = multiplies. */
int dnm_expm_chebyshev(dnm_mat *A, const void *x, void *y, int64_t n_local, double t, double tol,
                       const dnm_hooks *hooks, dnm_solver_stats *stats, void *stream);

/* The solvers keep their Krylov-basis allocation between calls and the reduced
 * density matrix its tile scratch; this frees them.  (The library keeps such
 * per-process buffers and small reduction scratch: calls are not re-entrant --
 * one host thread per process, as the reference's one thread per MPI rank.) */
int dnm_release_workspace(void);
/* bytes of Krylov basis currently cached (reusable by the next solve) */
int dnm_workspace_bytes(size_t *bytes);
/* Acquire (and touch: every page is written once) at least `bytes` of that workspace now, so that the next solve finds
 * it in place -- what a caller that times solves does once, untimed (bench.py: a first solve at L=30 otherwise pays
 * 1-2 s for 64 GiB of fresh device memory).  No-op when the cached workspace is already that large. */
int dnm_workspace_reserve(size_t bytes, void *stream);

/* ------------------------------------------------------------------ */
/* ranks -- the partitioned multiply as one native call                */
/* ------------------------------------------------------------------ */
/* The reference handles its ranks inside C: MatMult_CPU_Fast / _General post their scatters themselves
 * (bpetsc_template_2.c:413-504, 787-879), the CUDA shell all-gathers x (bcuda_template_2.cu:161-171), PETSc's
 * communicator comes with the Mat.  Here a dnm_comm is an RCCL communicator (one rank per GPU, xGMI) plus the
 * library's own exchange stream; RCCL is bound at run time (the library loads without it).
 *   dnm_comm_unique_id: ncclGetUniqueId -- one rank makes the 128 bytes, the host program hands them to the others
 *     (MPI_Bcast, a file, torch.distributed's store);
 *   dnm_comm_create: ncclCommInitRank -- collective;
 *   dnm_mat_mult_partitioned: y = A x on this rank's blocks, A built with the same (rank, nranks) partition: the
 *     exchange -- XOR-partner sub-blocks of a Full / Parity pair on 2^p ranks, the needed ranges of the column window
 *     of every other partition -- is posted on the exchange stream as one group of ncclSend / ncclRecv, the part of
 *     the multiply that reads nothing from other ranks runs under it on `stream`, the rest follows its event.
 *     Collective.
 *   dnm_mat_set_exchange: the scheme of a tiled Full / Parity operator on 2^p ranks.  DNM_EXCHANGE_TRANSPOSE (AUTO: from
 *     four ranks on, where an all-to-all puts less on the busiest xGMI link than the partner blocks): the operator is
 *     split inside the handle into the terms that flip no rank bit and the others rewritten for the layout in which the
 *     rank bits are local (the p index bits [f, f + p), f = n_local_bits - 1 - p, swapped with them); both parts are
 *     rank-local.  dnm_mat_mult_partitioned then sends the state through ONE all-to-all, runs the first part under it,
 *     the second on the redistributed state -- sub-piece by sub-piece as the pieces land, where the plan allows -- and
 *     adds what the returning all-to-all brings (bpetsc_template_2.c:866-873 overlaps assembly and compute block by
 *     block).  *chosen = the scheme in force: an operator that does not split (a term flips a rank bit and the field it
 *     would move to; too few local bits; no tiled plan) keeps DNM_EXCHANGE_PARTNER.  Call it before the first
 *     dnm_mat_mult_partitioned of A (or dnm_comm_forget(c, A) first: a communicator caches buffers per scheme), on every
 *     rank alike.  dnm_mat_exchange_parts hands out the two parts (owned by A; NULL under the partner scheme) and f, for
 *     hosts that run the schedule themselves.
 *   dnm_comm_allreduce: n doubles summed (op 0) / maximised (op 1) over the ranks, in place, host memory;
 *   dnm_comm_hooks: the dnm_hooks of dnm_expm_multiply / dnm_eigsolve filled with the two above (valid until the
 *     communicator is destroyed);
 *   dnm_comm_prepare: allocate now what the first dnm_mat_mult_partitioned of A would (receive buffers, the column
 *     window -- collective --, the two vectors of the transposed exchange), so that a solver sizing its Krylov basis
 *     to the free device memory sees what is really left;
 *   dnm_comm_forget: drop what the communicator caches for A (receive buffers, windows) before A is destroyed;
 *   dnm_comm_loopback (tests): a communicator of ONE rank stands for rank vrank of vranks; peer q's block of x is the
 *     device pointer peer_x[q] (peer_mat[q]: its handle, needed by window partitions and by the transposed exchange,
 *     whose returning pieces are what the PEERS computed -- the loop-back runs their second part too; without handles
 *     the returning pieces carry this rank's own data: right traffic, wrong numbers, for timing; entries for vrank
 *     itself are ignored) and every message becomes an RCCL send of this process to itself -- the schedules and the
 *     transport on a one-GPU box. */
typedef struct dnm_comm dnm_comm;
int dnm_comm_unique_id(void *id128);
int dnm_comm_create(const void *id128, int rank, int nranks, dnm_comm **out);
int dnm_comm_destroy(dnm_comm *c);
int dnm_comm_prepare(dnm_comm *c, dnm_mat *A, void *stream);
int dnm_comm_forget(dnm_comm *c, dnm_mat *A);
int dnm_comm_loopback(dnm_comm *c, int vrank, int vranks, const void *const *peer_x, dnm_mat *const *peer_mat);
int dnm_comm_allreduce(dnm_comm *c, double *vals, int n, int op);
/* Measurement: which half of the schedule the following dnm_mat_mult_partitioned calls run -- the whole multiply
 * (default), its messages alone (what the links sustain: nothing is computed, y is left alone), or its kernels alone (the
 * rank's compute with nothing on the links: results are not those of the multiply).  The three times together say how
 * much of the exchange the schedule hides (bench.py --gpus N).  The reference has no counterpart: its -log_view splits
 * VecScatterBegin/End from MatMult the same way. */
enum { DNM_PHASE_ALL = 0, DNM_PHASE_EXCHANGE = 1, DNM_PHASE_COMPUTE = 2 };
int dnm_comm_set_phase(dnm_comm *c, int phase);
int dnm_mat_mult_partitioned(dnm_mat *A, dnm_comm *c, const void *x, void *y, void *stream);
enum { DNM_EXCHANGE_AUTO = 0, DNM_EXCHANGE_PARTNER = 1, DNM_EXCHANGE_TRANSPOSE = 2 };
int dnm_mat_set_exchange(dnm_mat *A, int scheme, int *chosen);
int dnm_mat_exchange_parts(const dnm_mat *A, dnm_mat **lo, dnm_mat **hi, int *f);
/* The operator a handle holds (its own copies, as BuildContext_* keeps them): counts first (arrays may be NULL), then
 * masks[nmasks], mask_offsets[nmasks + 1], signs[nterms], coeffs[nterms] -- ONE double per term, the real part of the
 * coefficient where that is non-zero, else the imaginary part (ctx->real_coeffs, bpetsc_template_2.c:286-289; which of
 * the two it is follows from mask and sign: TERM_REAL, :398-404). */
int dnm_mat_operator(const dnm_mat *A, int64_t *nmasks, int64_t *nterms, int64_t *masks, int64_t *mask_offsets,
                     int64_t *signs, double *coeffs);
int dnm_comm_hooks(dnm_comm *c, dnm_mat *A, void *stream, dnm_hooks *out);

enum { DNM_WHICH_LOWEST = 0, DNM_WHICH_HIGHEST = 1, DNM_WHICH_EXTERIOR = 2 };

/* Thick-restart Lanczos (SLEPc EPSKRYLOVSCHUR on a HEP).  evals: [nev_max]
 * doubles; evecs (optional, may be NULL): device buffer of nev_max vectors of
 * n_local complex128 each, stride n_local.  nev_max >= nev.  tol<=0 -> 1e-8;
 * ncv<=0 -> max(2*nev, nev+15); max_its<=0 -> max(100, 2N/ncv).  Partial
 * re-orthogonalisation with a tolerance-driven trigger (DNM_EIGS_ORTHO=full:
 * every step against the whole basis, as SLEPc); stats->err_est returns the
 * measured largest relative residual of the returned pairs.
 * Automatic choices under default parameters (ncv <= 0) for large operators: nev = 1 -> Lanczos without a stored basis
 * (four work vectors); nev > 1 -> thick restart on a Chebyshev filter of H; ncv = -c (the caller's memory limit: at most
 * c vectors in all) with c < nev + 6 -> the pairs one after the other by the basis-free recurrence on the operator
 * deflated by the pairs found (four work vectors + the pairs; evals then count multiplicities -- a degenerate level
 * comes back once per copy, where a single Krylov space, SLEPc's included, shows one).  DESIGN.md section 5. */
int dnm_eigsolve(dnm_mat *A, int64_t n_local, int nev, int which, double tol,
                 int ncv, int max_its, uint64_t seed, const dnm_hooks *hooks,
                 int nev_max, double *evals, void *evecs,
                 dnm_solver_stats *stats, void *stream);

#ifdef __cplusplus
}
#endif
#endif
