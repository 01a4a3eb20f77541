/* A host written in plain C against include/dynamite_amd.h -- no Python, no torch: what a binding of the reference's
 * backend (src/dynamite/_backend/bpetsc.pyx:78-147) does, without the binding.
 *
 *   0.25 sum_i (XX + YY)_{i,i+1} on an open chain of L spins (MSC terms as operators.py:615-619 hands them over:
 *   XX = (mask 3 << i, sign 0, +0.25), YY = (mask 3 << i, sign 3 << i, -0.25)) on the Full space:
 *   dnm_mat_create, vectors from dnm_malloc, y = H x, <x, H y> = conj <y, H x>, dnm_eigsolve -- and the ground-state
 *   energy against the filled Fermi sea of the chain, the sum of the negative cos(pi j / (L + 1)).
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o c_abi_demo -Ldynamite_amd -ldynamite_amd -lm -Wl,-rpath,$PWD/dynamite_amd
 *   ./c_abi_demo [L]          (exit code 0: all checks passed; tests/test_gpu_c_host.py builds and runs it)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "dynamite_amd.h"

#define CK(call)                                                          \
  do {                                                                    \
    if ((call) != 0) {                                                    \
      fprintf(stderr, "%s failed: %s\n", #call, dnm_last_error());        \
      return 1;                                                           \
    }                                                                     \
  } while (0)

int main(int argc, char **argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 20;
  if (L < 12 || L > 30) { fprintf(stderr, "L in 12..30\n"); return 2; }
  const int64_t dim = (int64_t)1 << L;
  /* the operator: one mask per bond, two terms each */
  const int nmasks = L - 1;
  int64_t *masks = malloc(sizeof(int64_t) * nmasks), *offs = malloc(sizeof(int64_t) * (nmasks + 1));
  int64_t *signs = malloc(sizeof(int64_t) * 2 * nmasks);
  double *coeffs = malloc(sizeof(double) * 2 * 2 * nmasks);          /* complex128: (re, im) per term */
  for (int i = 0; i < nmasks; ++i) {
    masks[i] = (int64_t)3 << i;
    offs[i] = 2 * i;
    signs[2 * i] = 0;                  coeffs[4 * i] = 0.25;      coeffs[4 * i + 1] = 0.0;
    signs[2 * i + 1] = (int64_t)3 << i; coeffs[4 * i + 2] = -0.25; coeffs[4 * i + 3] = 0.0;
  }
  offs[nmasks] = 2 * nmasks;

  CK(dnm_set_device(0));
  dnm_subspace full = {0};
  full.type = DNM_FULL;
  full.L = L;
  dnm_mat *A = NULL;
  CK(dnm_mat_create(nmasks, masks, offs, signs, coeffs, &full, &full, 0, DNM_MAT_DEFAULT, NULL, &A));

  void *x, *y, *hx, *hy;
  CK(dnm_malloc(&x, 16 * (size_t)dim)); CK(dnm_malloc(&y, 16 * (size_t)dim));
  CK(dnm_malloc(&hx, 16 * (size_t)dim)); CK(dnm_malloc(&hy, 16 * (size_t)dim));
  CK(dnm_vec_set_random(x, dim, 1, 0, NULL));
  CK(dnm_vec_set_random(y, dim, 2, 0, NULL));
  CK(dnm_mat_mult(A, x, hx, NULL));
  CK(dnm_mat_mult(A, y, hy, NULL));
  double a[2], b[2], nx[2];
  CK(dnm_vec_dot(hy, x, dim, a, NULL));          /* sum (H y)_i conj(x_i) = <x, H y> */
  CK(dnm_vec_dot(y, hx, dim, b, NULL));          /* sum y_i conj((H x)_i) = <H x, y> */
  CK(dnm_vec_dot(x, x, dim, nx, NULL));
  CK(dnm_stream_synchronize(NULL));
  const double herm = hypot(a[0] - b[0], a[1] - b[1]) / nx[0];
  printf("L = %d: |<x, H y> - <H x, y>| / <x, x> = %.2e\n", L, herm);

  double exact = 0.0;
  for (int j = 1; j <= L; ++j) {
    const double e = cos(M_PI * j / (L + 1));
    if (e < 0) exact += e;
  }
  double evals[2] = {0, 0};
  dnm_solver_stats st;
  CK(dnm_eigsolve(A, dim, 1, DNM_WHICH_LOWEST, 1e-10, 0, 0, 7, NULL, 1, evals, NULL, &st, NULL));
  printf("ground-state energy %.12f, filled Fermi sea %.12f (%d multiplies, relative residual %.1e)\n", evals[0], exact,
         st.matvecs, st.err_est);

  CK(dnm_mat_destroy(A));
  CK(dnm_free(x)); CK(dnm_free(y)); CK(dnm_free(hx)); CK(dnm_free(hy));
  free(masks); free(offs); free(signs); free(coeffs);
  const int ok = herm < 1e-13 && fabs(evals[0] - exact) < 1e-8 * fabs(exact) && st.nconv >= 1;
  printf(ok ? "ok\n" : "FAILED\n");
  return ok ? 0 : 1;
}
