/*
 * dnm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See dnm_oracle.h for the rules on who may call this and how it is pinned.
 *
 * Every function names the reference location (relative to
 * /root/reference/src/dynamite/_backend/) whose algorithm it restates.
 */
#include "dnm_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int par64(orc_int v) { return __builtin_parityll((unsigned long long)v); }
static inline int pop64(orc_int v) { return __builtin_popcountll((unsigned long long)v); }
static inline int ctz64(orc_int v) { return __builtin_ctzll((unsigned long long)v); }

int orc_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_fill_test_vectors(orc_cplx *x, orc_cplx *y, orc_int n, int nthreads)
{
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
  for (orc_int i = 0; i < n; ++i) {
    x[i] = (double)(i % 1021 - 510) * (1.0 / 512) + I * ((double)(i % 509 - 254) * (1.0 / 256));
    y[i] = 0;
  }
}

/* ------------------------------------------------------------------ */
/* Subspace maps                                                      */
/* ------------------------------------------------------------------ */

/* Dim_*: bsubspace_impl.h:57-59 (Full), :112-114 (Parity), :187-189
 * (SpinConserve, read from the binomial table), :302-304 (Explicit). */
orc_int orc_dim(const orc_subspace *s)
{
  orc_int d = -1;
  switch (s->type) {
    case ORC_FULL:          d = (orc_int)1 << s->L; break;
    case ORC_PARITY:        d = (orc_int)1 << (s->L - 1); break;
    case ORC_SPIN_CONSERVE: d = s->nchoosek[s->k * s->ld_nchoosek + s->L]; break;
    case ORC_EXPLICIT:      d = s->dim; break;
  }
  if (s->xparity) d /= 2;   /* M /= 2; N /= 2;  bpetsc_template_2.c:227-230 */
  return d;
}

/* S2I_nocheck_SpinConserve, bsubspace_impl.h:191-202: colex rank, the j-th
 * set bit (counting from 1) at position n contributes C(n, j) when j <= n. */
static orc_int sc_rank(orc_int state, const orc_subspace *s)
{
  orc_int idx = 0, j = 0;
  while (state) {
    orc_int n = ctz64(state);
    ++j;
    if (j <= n) idx += s->nchoosek[j * s->ld_nchoosek + n];
    state &= state - 1;
  }
  return idx;
}

/* S2I_Explicit, bsubspace_impl.h:306-331: binary search over the sorted
 * states (right end starts at dim-1), optional index indirection. */
static orc_int ex_rank(orc_int state, const orc_subspace *s)
{
  orc_int lo = 0, hi = s->dim - 1;
  while (lo <= hi) {
    orc_int mid = (lo + hi) / 2;
    orc_int v = s->rmap_states[mid];
    if (v == state) return s->rmap_indices ? s->rmap_indices[mid] : mid;
    if (v < state) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

/* S2I_*: bsubspace_impl.h:61-63, :116-123, :204-208, :306-331. */
orc_int orc_s2i(orc_int state, const orc_subspace *s)
{
  switch (s->type) {
    case ORC_FULL:   return state;
    case ORC_PARITY: return (par64(state) == s->space) ? (state >> 1) : (orc_int)-1;
    case ORC_SPIN_CONSERVE:
      if (pop64(state) != s->k) return -1;
      return sc_rank(state, s);
    case ORC_EXPLICIT: return ex_rank(state, s);
  }
  return -1;
}

/* S2I_nocheck_*: bsubspace_impl.h:65-67, :125-127, :191-202.  (Explicit has
 * no nocheck form in the reference; the checked one is used.) */
orc_int orc_s2i_nocheck(orc_int state, const orc_subspace *s)
{
  switch (s->type) {
    case ORC_FULL:          return state;
    case ORC_PARITY:        return state >> 1;
    case ORC_SPIN_CONSERVE: return sc_rank(state, s);
    case ORC_EXPLICIT:      return ex_rank(state, s);
  }
  return -1;
}

/* I2S_*: bsubspace_impl.h:69-74, :129-134, :210-228 (greedy unranking from
 * bit L-1 down), :333-338. */
orc_int orc_i2s(orc_int idx, const orc_subspace *s)
{
  switch (s->type) {
    case ORC_FULL:   return idx;
    case ORC_PARITY: return (idx << 1) | (par64(idx) ^ s->space);
    case ORC_SPIN_CONSERVE: {
      orc_int state = 0, k = s->k;
      for (orc_int n = s->L; n > 0; --n) {
        orc_int here = (k > n - 1) ? 0 : s->nchoosek[k * s->ld_nchoosek + (n - 1)];
        state <<= 1;
        if (idx >= here) { idx -= here; --k; state |= 1; }
      }
      return state;
    }
    case ORC_EXPLICIT: return s->state_map[idx];
  }
  return -1;
}

/* NextState_*: bsubspace_impl.h:76-83, :136-143, :230-245 (next integer with
 * the same popcount), :340-347. */
orc_int orc_next_state(orc_int prev, orc_int idx, const orc_subspace *s)
{
  if (s->type != ORC_SPIN_CONSERVE) return orc_i2s(idx, s);
  int tz = ctz64(prev);
  prev >>= tz;
  prev += 1;
  int to = ctz64(prev);
  prev >>= to;
  prev <<= (to + tz);
  /* the reference writes (1 << (to-1)) - 1 with an int literal; to-1 < 63 */
  prev |= ((orc_int)1 << (to - 1)) - 1;
  return prev;
}

void orc_i2s_array(orc_int n, const orc_subspace *s, const orc_int *idxs, orc_int *states)
{
  for (orc_int i = 0; i < n; ++i) states[i] = orc_i2s(idxs[i], s);
}

void orc_s2i_array(orc_int n, const orc_subspace *s, const orc_int *states, orc_int *idxs)
{
  for (orc_int i = 0; i < n; ++i) idxs[i] = orc_s2i(states[i], s);
}

/* ------------------------------------------------------------------ */
/* Operator helpers                                                   */
/* ------------------------------------------------------------------ */

/* TERM_REAL, bpetsc_impl.h:34 */
static inline int term_is_real(orc_int mask, orc_int sign) { return !par64(mask & sign); }

/* bpetsc_template_2.c:286-290 */
void orc_real_coeffs(const orc_msc *msc, double *out)
{
  orc_int nterms = msc->mask_offsets[msc->nmasks];
  for (orc_int t = 0; t < nterms; ++t) {
    double re = creal(msc->coeffs[t]);
    out[t] = (re != 0) ? re : cimag(msc->coeffs[t]);
  }
}

/* bpetsc_template_1.c:169-202 */
int orc_precompute_diagonal(const orc_msc *msc, const orc_subspace *sub, double *diag)
{
  if (msc->nmasks == 0 || msc->masks[0] != 0) return 1;
  orc_int nterms = msc->mask_offsets[msc->nmasks];
  double *rc = (double *)malloc(sizeof(double) * (size_t)(nterms > 0 ? nterms : 1));
  orc_real_coeffs(msc, rc);
  orc_int dim = orc_dim(sub), state = 0;
  for (orc_int row = 0; row < dim; ++row) {
    state = (row == 0) ? orc_i2s(row, sub) : orc_next_state(state, row, sub);
    double v = 0;
    for (orc_int t = 0; t < msc->mask_offsets[1]; ++t) {
      int sgn = 1 - 2 * par64(state & msc->signs[t]);
      v += sgn * rc[t];
    }
    diag[row] = v;
  }
  free(rc);
  return 0;
}

/* ------------------------------------------------------------------ */
/* MatMult_CPU_General, single rank: bpetsc_template_2.c:371-412      */
/* ------------------------------------------------------------------ */
int orc_matvec_general(const orc_msc *msc, const orc_subspace *left,
                       const orc_subspace *right, const double *diag,
                       const orc_cplx *x, orc_cplx *b)
{
  orc_int nterms = msc->mask_offsets[msc->nmasks];
  double *rc = (double *)malloc(sizeof(double) * (size_t)(nterms > 0 ? nterms : 1));
  orc_real_coeffs(msc, rc);
  orc_int M = orc_dim(left), ket = 0;
  for (orc_int row = 0; row < M; ++row) b[row] = 0;           /* VecSet(b,0), :364 */
  for (orc_int row = 0; row < M; ++row) {
    ket = (row == 0) ? orc_i2s(row, left) : orc_next_state(ket, row, left);
    orc_int mi = 0;
    if (diag) { b[row] += diag[row] * x[row]; mi = 1; }       /* :382-387 */
    for (; mi < msc->nmasks; ++mi) {
      orc_int bra = ket ^ msc->masks[mi];
      orc_int col = orc_s2i(bra, right);
      if (col == -1) continue;                                /* projection, :393-395 */
      orc_cplx val = 0;
      for (orc_int t = msc->mask_offsets[mi]; t < msc->mask_offsets[mi + 1]; ++t) {
        int sgn = 1 - 2 * par64(bra & msc->signs[t]);         /* sign on the column state */
        if (term_is_real(msc->masks[mi], msc->signs[t])) val += sgn * rc[t];
        else val += I * sgn * rc[t];
      }
      b[row] += val * x[col];
    }
  }
  free(rc);
  return 0;
}

/* ------------------------------------------------------------------ */
/* MatMult_CPU_Fast: bpetsc_template_2.c:713-889 + helpers :575-683   */
/* ------------------------------------------------------------------ */
#define BLK   ((orc_int)1 << 11)   /* VECSET_CACHE_SIZE, :524 */
#define LKP   ((orc_int)1 << 6)    /* LKP_SIZE, :525 */
#define LKPM  (LKP - 1)

/* compute_sign_lookup (:575-584) and compute_parity_sign_lookup (:586-596):
 * tab[i*LKP + j] = +1/-1 from parity(i&j) [^ parity(j) ^ space]. */
static void build_lookup(orc_int *tab, int with_parity, orc_int space)
{
  for (orc_int i = 0; i < LKP; ++i)
    for (orc_int j = 0; j < LKP; ++j) {
      orc_int p = par64(i & j);
      if (with_parity) p ^= par64(j) ^ space;
      tab[i * LKP + j] = p ? -1 : 1;
    }
}

/* sum_term (:637-683): add +-coeff (as a real or an imaginary number) into the per-row coefficient buffer.
 * The sign of row `block_start + c` under sign mask `s` splits into a parity of the bits above the low 6
 * (constant over a run of 64 rows) and a 64x64 lookup on the low 6; check_parity adds the Parity-subspace
 * dropped-bit correction.  As in the reference the loop is unswitched by hand over (real | imaginary) x
 * (lookup needed | sign mask clear on the low 6 bits | parity lookup): six loop nests whose inner loop over 64
 * rows has no branch, so this leg is a fair stand-in for the reference's cost per term. */
#define ORC_SUM_LOOP(SIGN_OF_J, ADD_STMT, WITH_PARITY)                          \
  for (orc_int c = 0; c < BLK; c += LKP) {                                      \
    const orc_int hi = (c + block_start) & ~LKPM;                               \
    int flip = par64(hi & s);                                                   \
    if (WITH_PARITY) flip ^= par64(hi);                                         \
    const double v = flip ? -coeff : coeff;                                     \
    orc_cplx *a = acc + c;                                                      \
    for (orc_int j = 0; j < LKP; ++j) {                                         \
      const double w = (double)(SIGN_OF_J) * v;                                 \
      ADD_STMT;                                                                 \
    }                                                                           \
  }

static void add_term(orc_int block_start, orc_int s, int is_real, double coeff,
                     int check_parity, const orc_int *tab, orc_cplx *acc)
{
  const orc_int *row = tab + (s & LKPM) * LKP;
  if (check_parity) {
    if (is_real) { ORC_SUM_LOOP(row[j], a[j] += w, 1) }
    else         { ORC_SUM_LOOP(row[j], a[j] += I * w, 1) }
  } else if (s & LKPM) {
    if (is_real) { ORC_SUM_LOOP(row[j], a[j] += w, 0) }
    else         { ORC_SUM_LOOP(row[j], a[j] += I * w, 0) }
  } else {
    if (is_real) { ORC_SUM_LOOP(1, a[j] += w, 0) }
    else         { ORC_SUM_LOOP(1, a[j] += I * w, 0) }
  }
}
#undef ORC_SUM_LOOP

/* do_cache_product (:598-635): values[c] += acc[c] * x[(block_start + c) ^ m].  XOR with m maps an aligned run
 * of 2^ctz(m) consecutive rows onto consecutive columns, so for runs of at least ITER_CUTOFF (8, :515) the
 * reference walks run by run with unit-stride inner loops; shorter runs take the element-wise loop. */
#define ITER_CUTOFF 8
static void apply_mask(orc_int m, orc_int block_start, orc_int x_start,
                       const orc_cplx *acc, const orc_cplx *xloc, orc_cplx *vals)
{
  const orc_int run = m ? ((orc_int)1 << ctz64(m)) : BLK;
  if (run < ITER_CUTOFF) {
    for (orc_int c = 0; c < BLK; ++c) {
      orc_int r = (block_start + c) ^ m;
      vals[c] += acc[c] * xloc[r - x_start];
    }
    return;
  }
  for (orc_int c = 0; c < BLK;) {
    const orc_int r = (block_start + c) ^ m;
    orc_int stop = run - (r % run);
    if (stop > BLK - c) stop = BLK - c;
    const orc_cplx *xs = xloc + (r - x_start);
    for (orc_int j = 0; j < stop; ++j) vals[c + j] += acc[c + j] * xs[j];
    c += stop;
  }
}

/* One 2^11-row block of the loop body at :799-874.  mask range [m0, m1).
 * use_diag mirrors `proc_idx==0 && ctx->diag` (:811-818).  out receives the
 * block (ADD_VALUES into b, :870). */
static void fast_block(const orc_msc *msc, const double *rc, const orc_subspace *sub,
                       const orc_int *tab, const orc_int *ptab,
                       orc_int block_start, orc_int m0, orc_int m1,
                       const double *diag_loc, orc_int x_start, const orc_cplx *xloc,
                       orc_cplx *acc, orc_cplx *vals, orc_cplx *out)
{
  memset(vals, 0, sizeof(orc_cplx) * (size_t)BLK);
  memset(acc, 0, sizeof(orc_cplx) * (size_t)BLK);
  orc_int mi = m0;
  if (diag_loc) {
    for (orc_int c = 0; c < BLK; ++c)
      vals[c] = diag_loc[(block_start - x_start) + c] * xloc[(block_start - x_start) + c];
    mi = 1;
  }
  for (; mi < m1; ++mi) {
    orc_int mask = msc->masks[mi];
    if (sub->type == ORC_PARITY && par64(mask)) continue;     /* :822-827 */
    orc_int m = orc_s2i_nocheck(mask, sub);
    for (orc_int t = msc->mask_offsets[mi]; t < msc->mask_offsets[mi + 1]; ++t) {
      orc_int sg = msc->signs[t];
      orc_int s = orc_s2i_nocheck(sg, sub);
      double c = par64(mask & sg) ? -rc[t] : rc[t];           /* :844-845 */
      int r = term_is_real(mask, sg);
      if (sub->type == ORC_PARITY && (sg & 1)) add_term(block_start, s, r, c, 1, ptab, acc);
      else add_term(block_start, s, r, c, 0, tab, acc);
    }
    apply_mask(m, block_start, x_start, acc, xloc, vals);
    memset(acc, 0, sizeof(orc_cplx) * (size_t)BLK);
  }
  for (orc_int c = 0; c < BLK; ++c) out[c] += vals[c];
}

static int fast_supported(const orc_subspace *sub, orc_int dim_local)
{
  if (sub->type != ORC_FULL && sub->type != ORC_PARITY) return 0;
  return dim_local > BLK && (dim_local & (dim_local - 1)) == 0;
}

int orc_matvec_fast(const orc_msc *msc, const orc_subspace *sub,
                    const double *diag, const orc_cplx *x, orc_cplx *b,
                    int nthreads)
{
  orc_int dim = orc_dim(sub);
  if (!fast_supported(sub, dim)) return 1;
  orc_int nterms = msc->mask_offsets[msc->nmasks];
  double *rc = (double *)malloc(sizeof(double) * (size_t)(nterms > 0 ? nterms : 1));
  orc_real_coeffs(msc, rc);
  orc_int *tab = (orc_int *)malloc(sizeof(orc_int) * LKP * LKP);
  orc_int *ptab = (orc_int *)malloc(sizeof(orc_int) * LKP * LKP);
  build_lookup(tab, 0, 0);
  build_lookup(ptab, 1, sub->space);
  orc_int nblk = dim / BLK;
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    orc_cplx *acc = (orc_cplx *)malloc(sizeof(orc_cplx) * (size_t)BLK);
    orc_cplx *vals = (orc_cplx *)malloc(sizeof(orc_cplx) * (size_t)BLK);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (orc_int kb = 0; kb < nblk; ++kb) {
      orc_int bs = kb * BLK;
      memset(b + bs, 0, sizeof(orc_cplx) * (size_t)BLK);      /* VecSet(b,0), :749 */
      fast_block(msc, rc, sub, tab, ptab, bs, 0, msc->nmasks, diag, 0, x,
                 acc, vals, b + bs);
    }
    free(acc); free(vals);
  }
  free(rc); free(tab); free(ptab);
  return 0;
}

int orc_matvec(const orc_msc *msc, const orc_subspace *left,
               const orc_subspace *right, const double *diag,
               const orc_cplx *x, orc_cplx *b, int nthreads)
{
  /* :513, :542-552 */
  int same = left->type == right->type &&
             (left->type == ORC_FULL ||
              (left->type == ORC_PARITY && left->space == right->space)) &&
             left->L == right->L;
  if (same && fast_supported(left, orc_dim(left)))
    return orc_matvec_fast(msc, left, diag, x, b, nthreads);
  return orc_matvec_general(msc, left, right, diag, x, b);
}

/* compute_mask_starts (:686-709): first mask whose index-space image has the
 * process prefix p, masks being sorted. */
static void mask_starts(const orc_msc *msc, const orc_subspace *sub,
                        orc_int n_local_spins, int P, orc_int *starts)
{
  orc_int mi = 0;
  for (int p = 0; p < P; ++p) {
    while (mi < msc->nmasks &&
           orc_s2i_nocheck(msc->masks[mi], sub) < ((orc_int)p << n_local_spins)) ++mi;
    starts[p] = mi;
  }
  starts[P] = msc->nmasks;
}

int orc_matvec_fast_ranks(const orc_msc *msc, const orc_subspace *sub,
                          const double *diag, const orc_cplx *x, orc_cplx *b,
                          int P)
{
  orc_int dim = orc_dim(sub);
  if (P < 1 || (P & (P - 1))) return 1;                       /* :742-744 */
  orc_int nloc = dim / P;
  if (!fast_supported(sub, nloc)) return 1;
  orc_int nterms = msc->mask_offsets[msc->nmasks];
  double *rc = (double *)malloc(sizeof(double) * (size_t)(nterms > 0 ? nterms : 1));
  orc_real_coeffs(msc, rc);
  orc_int *tab = (orc_int *)malloc(sizeof(orc_int) * LKP * LKP);
  orc_int *ptab = (orc_int *)malloc(sizeof(orc_int) * LKP * LKP);
  build_lookup(tab, 0, 0);
  build_lookup(ptab, 1, sub->space);
  orc_cplx *acc = (orc_cplx *)malloc(sizeof(orc_cplx) * (size_t)BLK);
  orc_cplx *vals = (orc_cplx *)malloc(sizeof(orc_cplx) * (size_t)BLK);
  orc_int *starts = (orc_int *)malloc(sizeof(orc_int) * (size_t)(P + 1));
  orc_int nls = ctz64(nloc);                                  /* :770 */
  mask_starts(msc, sub, nls, P, starts);
  orc_int proc_mask = ~(((orc_int)1 << nls) - 1);             /* (-1) << n_local_spins, :781 */
  for (orc_int r = 0; r < dim; ++r) b[r] = 0;
  for (int me = 0; me < P; ++me) {
    orc_int x_start = (orc_int)me * nloc;
    orc_int proc_me = (orc_int)me << nls;
    for (int p = 0; p < P; ++p) {
      if (starts[p] == starts[p + 1]) continue;               /* :790 */
      if (starts[p] == msc->nmasks) break;                    /* :793 */
      orc_int m = orc_s2i_nocheck(msc->masks[starts[p]], sub);
      orc_int target = proc_mask & (proc_me ^ m);             /* :796-797 */
      for (orc_int bs = target; bs < target + nloc; bs += BLK) {
        const double *dl = (p == 0 && diag) ? diag + x_start : NULL;
        fast_block(msc, rc, sub, tab, ptab, bs, starts[p], starts[p + 1], dl,
                   x_start, x + x_start, acc, vals, b + bs);
      }
    }
  }
  free(rc); free(tab); free(ptab); free(acc); free(vals); free(starts);
  return 0;
}

/* ------------------------------------------------------------------ */
/* MatNorm_CPU (NORM_INFINITY): bpetsc_template_2.c:906-981           */
/* ------------------------------------------------------------------ */
int orc_infnorm(const orc_msc *msc, const orc_subspace *left,
                const orc_subspace *right, double *nrm)
{
  orc_int nterms = msc->mask_offsets[msc->nmasks];
  double *rc = (double *)malloc(sizeof(double) * (size_t)(nterms > 0 ? nterms : 1));
  orc_real_coeffs(msc, rc);
  orc_int M = orc_dim(left);
  double best = 0;
  for (orc_int row = 0; row < M; ++row) {
    orc_int ket = orc_i2s(row, left);
    double sum = 0, err = 0;                                  /* Kahan, :964-967 */
    for (orc_int mi = 0; mi < msc->nmasks; ++mi) {
      orc_int bra = ket ^ msc->masks[mi];
      if (orc_s2i(bra, right) == -1) continue;
      orc_cplx cs = 0;
      for (orc_int t = msc->mask_offsets[mi]; t < msc->mask_offsets[mi + 1]; ++t) {
        int sgn = 1 - 2 * par64(bra & msc->signs[t]);
        if (term_is_real(msc->masks[mi], msc->signs[t])) cs += sgn * rc[t];
        else cs += I * (sgn * rc[t]);
      }
      double comp = cabs(cs) - err;
      double tot = sum + comp;
      err = (tot - sum) - comp;
      sum = tot;
    }
    if (sum > best) best = sum;
  }
  *nrm = best;
  free(rc);
  return 0;
}

/* ------------------------------------------------------------------ */
/* CheckConserves: bpetsc_template_2.c:990-1056 (single rank)         */
/* ------------------------------------------------------------------ */
int orc_check_conserves(const orc_msc *msc, const orc_subspace *left,
                        const orc_subspace *right, int *result)
{
  orc_int N = orc_dim(right);
  int ok = 1;
  for (orc_int col = 0; col < N && ok; ++col) {
    orc_int bra = orc_i2s(col, right);
    for (orc_int mi = 0; mi < msc->nmasks; ++mi) {
      orc_int ket = bra ^ msc->masks[mi];
      if (orc_s2i(ket, left) != -1) continue;
      orc_cplx v = 0;
      for (orc_int t = msc->mask_offsets[mi]; t < msc->mask_offsets[mi + 1]; ++t) {
        int sgn = 1 - 2 * par64(bra & msc->signs[t]);
        v += sgn * msc->coeffs[t];
      }
      if (v != 0) { ok = 0; break; }
    }
  }
  *result = ok;
  return 0;
}


/* ------------------------------------------------------------------ */
/* Reduced density matrix (bpetsc_template_1.c:15-165)                */
/* ------------------------------------------------------------------ */

/* combine_states, :30-56: interleave the kept and the traced bits (keep sorted) */
static orc_int rdm_combine(orc_int keep_state, orc_int tr_state, const orc_int *keep,
                           orc_int keep_size, orc_int L)
{
  orc_int rtn = 0, keep_idx = 0, tr_idx = 0;
  for (orc_int pos = 0; pos < L; ++pos) {
    orc_int bit;
    if (keep_idx < keep_size && pos == keep[keep_idx]) {
      bit = (keep_state >> keep_idx) & 1;
      ++keep_idx;
    } else {
      bit = (tr_state >> tr_idx) & 1;
      ++tr_idx;
    }
    rtn |= bit << pos;
  }
  return rtn;
}

int orc_rdm(const orc_subspace *sub, const orc_cplx *x, orc_int keep_size,
            const orc_int *keep, orc_cplx *rtn)
{
  for (orc_int i = 1; i < keep_size; ++i)
    if (keep[i] <= keep[i - 1]) return 1;                       /* :117-121 */
  orc_int kd = (orc_int)1 << keep_size;
  orc_int *st = (orc_int *)malloc(sizeof(orc_int) * (size_t)kd);
  orc_cplx *cv = (orc_cplx *)malloc(sizeof(orc_cplx) * (size_t)kd);
  memset(rtn, 0, sizeof(orc_cplx) * (size_t)kd * (size_t)kd);
  orc_int tr_dim = (orc_int)1 << (sub->L - keep_size);          /* :150 */
  for (orc_int tr = 0; tr < tr_dim; ++tr) {
    orc_int nf = 0;                                             /* fill_combine_array, :58-85 */
    for (orc_int ks = 0; ks < kd; ++ks) {
      orc_int idx = orc_s2i(rdm_combine(ks, tr, keep, keep_size, sub->L), sub);
      if (idx != -1) { st[nf] = ks; cv[nf] = x[idx]; ++nf; }
    }
    for (orc_int i = 0; i < nf; ++i) {                          /* :153-159 */
      orc_int off = st[i] * kd;
      orc_cplx a = cv[i];
      for (orc_int j = 0; j < nf; ++j) rtn[off + st[j]] += a * conj(cv[j]);
    }
  }
  free(st);
  free(cv);
  return 0;
}
