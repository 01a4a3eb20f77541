/*
 * dnm_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's CPU algorithm for the matrix-free
 * H|psi> path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call into this library, and only as the checker /
 * the reported CPU baseline -- never as the thing shipped or measured as the
 * product.
 *
 * Parity pinning: checked against (i) the known-answer tables of the
 * reference's own unit tests (tests/golden/known_answers.json, transcribed
 * from tests/unit/test_msc_tools.py and tests/unit/test_subspaces.py) and
 * (ii) golden vectors produced by importing the reference's Python layer
 * (msc_tools.msc_to_numpy, the format-defining builder) in the build
 * container: tests/golden/make_golden.py -> tests/golden/ (npz files).
 *
 * The reference C sources themselves are NOT buildable here: every file
 * under src/dynamite/_backend includes <petsc.h> (PETSc 3.20.5), which the
 * image lacks, so there is no oracle/_ref build.
 *
 * All integers are 64-bit (the reference's PetscInt under
 * --with-64-bit-indices, bbuild.pyx:28-33); scalars are C99 double _Complex
 * (PetscScalar in a complex build).
 */
#ifndef DNM_ORACLE_H
#define DNM_ORACLE_H

#include <stdint.h>
#include <complex.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t orc_int;
typedef double _Complex orc_cplx;

/* bsubspace_impl.h:17-23 */
enum { ORC_FULL = 0, ORC_PARITY = 1, ORC_EXPLICIT = 2, ORC_SPIN_CONSERVE = 3 };

/* One descriptor for all four subspace kinds (data_Full/Parity/SpinConserve/
 * Explicit of bsubspace_impl.h:41-44,95-99,161-167,265-272 folded together). */
typedef struct {
  int type;
  orc_int L;
  orc_int space;            /* Parity: 0 even / 1 odd */
  orc_int k;                /* SpinConserve */
  orc_int ld_nchoosek;      /* SpinConserve: L+1 */
  const orc_int *nchoosek;  /* SpinConserve: (k+1) x (L+1), nchoosek[kk*ld+LL] = C(LL,kk) */
  orc_int dim;              /* Explicit */
  const orc_int *state_map;    /* Explicit: idx -> state */
  const orc_int *rmap_indices; /* Explicit: NULL if state_map is sorted */
  const orc_int *rmap_states;  /* Explicit: sorted states */
  /* XParity (subspaces.py:532-800): the operator has been rewritten by
   * XParity.reduce_msc and the basis is the first half of the parent's, so
   * "the only thing we have to do in the backend" is halve the dimension
   * (bpetsc_template_2.c:78-85,223-230,1005-1008). */
  orc_int xparity;
} orc_subspace;

orc_int orc_dim(const orc_subspace *s);
orc_int orc_i2s(orc_int idx, const orc_subspace *s);
orc_int orc_s2i(orc_int state, const orc_subspace *s);
orc_int orc_s2i_nocheck(orc_int state, const orc_subspace *s);
orc_int orc_next_state(orc_int prev_state, orc_int idx, const orc_subspace *s);
void orc_i2s_array(orc_int n, const orc_subspace *s, const orc_int *idxs, orc_int *states);
void orc_s2i_array(orc_int n, const orc_subspace *s, const orc_int *states, orc_int *idxs);

/* The operator in the reference's marshalled form (shell_context.h:4-10,
 * operators.py:653-669): sorted unique masks, CSR-like offsets into
 * signs/coeffs. */
typedef struct {
  orc_int nmasks;
  const orc_int *masks;
  const orc_int *mask_offsets;
  const orc_int *signs;
  const orc_cplx *coeffs;
} orc_msc;

/* BuildContext_CPU's folding of complex coefficients to one double each
 * (bpetsc_template_2.c:286-290).  out has mask_offsets[nmasks] entries. */
void orc_real_coeffs(const orc_msc *msc, double *out);

/* PrecomputeDiagonal_CPU (bpetsc_template_1.c:169-202).  Returns 0 and fills
 * diag[dim] if masks[0]==0, returns 1 (no diagonal) otherwise. */
int orc_precompute_diagonal(const orc_msc *msc, const orc_subspace *sub, double *diag);

/* MatMult_CPU_General, single-rank branch (bpetsc_template_2.c:371-412).
 * diag may be NULL.  b is overwritten. */
int orc_matvec_general(const orc_msc *msc, const orc_subspace *left,
                       const orc_subspace *right, const double *diag,
                       const orc_cplx *x, orc_cplx *b);

/* MatMult_CPU_Fast, single-rank (bpetsc_template_2.c:713-889 with helpers
 * :575-683), Full/Full or Parity/Parity(same space) only; requires
 * dim > 2^11 and a power of two.  nthreads>1 distributes the independent
 * 2^11-row blocks over OpenMP threads (the reference distributes them over
 * MPI ranks); nthreads<=1 is the scalar single-rank loop. */
int orc_matvec_fast(const orc_msc *msc, const orc_subspace *sub,
                    const double *diag, const orc_cplx *x, orc_cplx *b,
                    int nthreads);

/* MatMult_CPU dispatch (bpetsc_template_2.c:530-561): Fast iff same Full or
 * Parity(same space) subspace and local_size > 2^11, else General. */
int orc_matvec(const orc_msc *msc, const orc_subspace *left,
               const orc_subspace *right, const double *diag,
               const orc_cplx *x, orc_cplx *b, int nthreads);

/* MatMult_CPU_Fast as P MPI ranks would run it (bpetsc_template_2.c:787-879):
 * rank p owns x[p*dim/P .. (p+1)*dim/P) and computes, for every XOR-partner
 * process, blocks of the PARTNER's output rows from its local x.  Emulated
 * serially; used to pin the multi-GPU partition logic.  P power of two. */
int orc_matvec_fast_ranks(const orc_msc *msc, const orc_subspace *sub,
                          const double *diag, const orc_cplx *x, orc_cplx *b,
                          int P);

/* MatNorm_CPU, NORM_INFINITY with Kahan row sums (bpetsc_template_2.c:906-981). */
int orc_infnorm(const orc_msc *msc, const orc_subspace *left,
                const orc_subspace *right, double *nrm);

/* CheckConserves (bpetsc_template_2.c:990-1056), single rank, xparity=0. */
int orc_check_conserves(const orc_msc *msc, const orc_subspace *left,
                        const orc_subspace *right, int *result);

int orc_max_threads(void);
/* bench support: x[i] = ((i % 1021) - 510)/512 + i ((i % 509) - 254)/256 and y = 0, written by the threads that
 * will later work on those blocks (first touch places the pages on their NUMA nodes, as rank-local PETSc vectors
 * would be) */
void orc_fill_test_vectors(orc_cplx *x, orc_cplx *y, orc_int n, int nthreads);

#ifdef __cplusplus
}
#endif

/* rdm_<SUBSPACE> (bpetsc_template_1.c:15-165): reduced density matrix of the state x
 * on the spins keep[0] < keep[1] < ... (bit i of a kept configuration is spin keep[i],
 * reduce_state :15-28).  rtn is (2^keep_size)^2 row-major, rtn[i*dim+j] = sum over the
 * traced configurations of psi(i,tr) conj(psi(j,tr)), psi = 0 outside the subspace
 * (fill_combine_array :58-85).  Returns 1 if keep is not strictly increasing (:117-121). */
int orc_rdm(const orc_subspace *sub, const orc_cplx *x, orc_int keep_size,
            const orc_int *keep, orc_cplx *rtn);

#endif
