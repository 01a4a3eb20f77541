// Host-side helpers shared by the vector C ABI and the Krylov drivers.
#pragma once

#include <cmath>

#include "kernels.h"
#include "mat.h"

namespace dnm {

int vec_scratch(size_t bytes, double **p);
int vec_upload_coefs(const double *host, size_t ndoubles, hipStream_t st, const double **dev);
// h_host[2*j], h_host[2*j+1] = V_j^H w ; synchronises the stream
int vec_mdot_host(const void *V, int64_t ldv, int nv, const void *w, int64_t n, double *h_host,
                  hipStream_t st);

// p -= a v + b u in one sweep; *norm2_host = |p|^2 afterwards (local part); synchronises the stream
// p = scale * (p - (are + i aim) v - b u); *norm2_host = |p|^2 of the result (this rank's part)
int vec_lanczos_update_host(void *p, const void *v, const void *u, int64_t n, double are, double aim, double b,
                            double *norm2_host, hipStream_t st, double scale = 1.0);
// y -= b z (z may be null); out3_host = { Re <x,y>, Im <x,y>, |y|^2 } (this rank's part)
int vec_lanczos_dot_host(void *y, const void *z, const void *x, int64_t n, double b, double *out3_host,
                         hipStream_t st, double yscale = 1.0);

}  // namespace dnm
