// Krylov drivers on top of the matrix-free multiply and the fused vector
// kernels.  They replace the two SLEPc solvers dynamite calls
// (src/dynamite/computations.py:89-112 MFN "expokit", :208-257 EPS
// Krylov-Schur on a Hermitian problem).  SLEPc is a third-party dependency
// that is not part of the reference tree (slepc4py == 3.20.2,
// pyproject.toml:24): the algorithms below restate the published methods
// (Sidje, "Expokit", ACM TOMS 24 (1998) -- the scheme SLEPc's MFNEXPOKIT
// implements; Wu & Simon thick-restart Lanczos = Krylov-Schur for Hermitian
// matrices, Stewart 2001).  Step counts / step sizes are not pinned by any
// reference test; results are (tests/integration/test_evolve.py,
// test_eigsolve.py tolerances).
//
// The host only handles (m+2)^2 dense matrices; every O(dim) operation is a
// HIP kernel.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <vector>

#include "vec_api.h"

namespace dnm {

typedef std::complex<double> zc;

// ---------------------------------------------------------------------------
// small dense helpers (column-major, leading dimension = n)
// ---------------------------------------------------------------------------
static void zgemm(int n, const std::vector<zc> &A, const std::vector<zc> &B, std::vector<zc> &C) {
  C.assign((size_t)n * n, zc(0));
  for (int j = 0; j < n; ++j)
    for (int k = 0; k < n; ++k) {
      const zc b = B[(size_t)j * n + k];
      if (b == zc(0)) continue;
      for (int i = 0; i < n; ++i) C[(size_t)j * n + i] += A[(size_t)k * n + i] * b;
    }
}

// solve A X = B in place (B overwritten by X), partial pivoting; A destroyed
static int zsolve(int n, std::vector<zc> &A, std::vector<zc> &Bm) {
  for (int c = 0; c < n; ++c) {
    int piv = c;
    double best = std::abs(A[(size_t)c * n + c]);
    for (int r = c + 1; r < n; ++r)
      if (std::abs(A[(size_t)c * n + r]) > best) { best = std::abs(A[(size_t)c * n + r]); piv = r; }
    if (best == 0.0) return 1;
    if (piv != c) {
      for (int j = 0; j < n; ++j) {
        std::swap(A[(size_t)j * n + c], A[(size_t)j * n + piv]);
        std::swap(Bm[(size_t)j * n + c], Bm[(size_t)j * n + piv]);
      }
    }
    const zc inv = zc(1) / A[(size_t)c * n + c];
    for (int r = c + 1; r < n; ++r) {
      const zc f = A[(size_t)c * n + r] * inv;
      if (f == zc(0)) continue;
      for (int j = c; j < n; ++j) A[(size_t)j * n + r] -= f * A[(size_t)j * n + c];
      for (int j = 0; j < n; ++j) Bm[(size_t)j * n + r] -= f * Bm[(size_t)j * n + c];
    }
  }
  for (int j = 0; j < n; ++j)
    for (int r = n - 1; r >= 0; --r) {
      zc s = Bm[(size_t)j * n + r];
      for (int k = r + 1; k < n; ++k) s -= A[(size_t)k * n + r] * Bm[(size_t)j * n + k];
      Bm[(size_t)j * n + r] = s / A[(size_t)r * n + r];
    }
  return 0;
}

// exp(A) by scaling and squaring with the diagonal Pade approximant of degree
// 13 (Higham 2005 coefficients), complex dense.
static int zexpm(int n, const std::vector<zc> &Ain, std::vector<zc> &E) {
  static const double b[14] = {64764752532480000., 32382376266240000., 7771770303897600.,
                               1187353796428800.,  129060195264000.,   10559470521600.,
                               670442572800.,      33522128640.,       1323241920.,
                               40840800.,          960960.,            16380.,
                               182.,               1.};
  double nrm = 0;
  for (int j = 0; j < n; ++j) {
    double cs = 0;
    for (int i = 0; i < n; ++i) cs += std::abs(Ain[(size_t)j * n + i]);
    nrm = std::max(nrm, cs);
  }
  int s = 0;
  const double theta13 = 5.371920351148152;
  if (nrm > theta13) s = std::max(0, (int)std::ceil(std::log2(nrm / theta13)));
  std::vector<zc> A = Ain;
  const double sc = std::ldexp(1.0, -s);
  for (auto &v : A) v *= sc;
  std::vector<zc> A2, A4, A6, U, V, T1, T2;
  zgemm(n, A, A, A2);
  zgemm(n, A2, A2, A4);
  zgemm(n, A4, A2, A6);
  const size_t nn = (size_t)n * n;
  // U = A [A6 (b13 A6 + b11 A4 + b9 A2) + b7 A6 + b5 A4 + b3 A2 + b1 I]
  T1.assign(nn, zc(0));
  for (size_t i = 0; i < nn; ++i) T1[i] = b[13] * A6[i] + b[11] * A4[i] + b[9] * A2[i];
  zgemm(n, A6, T1, T2);
  for (size_t i = 0; i < nn; ++i) T2[i] += b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i];
  for (int i = 0; i < n; ++i) T2[(size_t)i * n + i] += b[1];
  zgemm(n, A, T2, U);
  // V = A6 (b12 A6 + b10 A4 + b8 A2) + b6 A6 + b4 A4 + b2 A2 + b0 I
  for (size_t i = 0; i < nn; ++i) T1[i] = b[12] * A6[i] + b[10] * A4[i] + b[8] * A2[i];
  zgemm(n, A6, T1, V);
  for (size_t i = 0; i < nn; ++i) V[i] += b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i];
  for (int i = 0; i < n; ++i) V[(size_t)i * n + i] += b[0];
  // (V - U) E = (V + U)
  std::vector<zc> P(nn), Q(nn);
  for (size_t i = 0; i < nn; ++i) { P[i] = V[i] + U[i]; Q[i] = V[i] - U[i]; }
  if (zsolve(n, Q, P)) return 1;
  E.swap(P);
  for (int k = 0; k < s; ++k) {
    zgemm(n, E, E, T1);
    E.swap(T1);
  }
  return 0;
}

// cyclic Jacobi for a dense real symmetric matrix: A = S diag(w) S^T
static void jacobi_eig(int n, std::vector<double> &A, std::vector<double> &w, std::vector<double> &Sv) {
  Sv.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) Sv[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        if (i != j) off += A[(size_t)j * n + i] * A[(size_t)j * n + i];
        else diag += A[(size_t)j * n + i] * A[(size_t)j * n + i];
      }
    if (off <= 1e-32 * (diag + 1e-300)) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[(size_t)q * n + p];
        if (apq == 0.0) continue;
        const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < n; ++k) {   // columns p, q
          const double akp = A[(size_t)p * n + k], akq = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = c * akp - s * akq;
          A[(size_t)q * n + k] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {   // rows p, q
          const double apk = A[(size_t)k * n + p], aqk = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = c * apk - s * aqk;
          A[(size_t)k * n + q] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double skp = Sv[(size_t)p * n + k], skq = Sv[(size_t)q * n + k];
          Sv[(size_t)p * n + k] = c * skp - s * skq;
          Sv[(size_t)q * n + k] = s * skp + c * skq;
        }
      }
  }
  w.resize(n);
  for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
}

// ---------------------------------------------------------------------------
// distributed plumbing
// ---------------------------------------------------------------------------
// Chebyshev filter p(A) = T_d((A - c) / h) / T_d((ref - c) / h): on the interval [c - h, c + h] |p| <= 1 / |T_d(ref)|,
// outside it p grows exponentially with the distance -- Lanczos on p(A) sees the few eigenvalues beyond one end of
// the interval as huge, well separated ones.  Evaluated by the three-term recurrence on unnormalised vectors
// s_j = h^j T_j: s_{j+1} = 2 (A - c) s_j - h^2 s_{j-1} (s_1 = (A - c) s_0), every term ONE fused multiply
// y = A x - b z + c2 z2 (dnm_mat_mult_sub2); the scale that brings s_d back to O(1) is known on the host.
// seeded start vector in the layout of A's vectors (padding of an internal SpinConserve layout stays zero)
static int random_start(dnm_mat *A, void *x, int64_t n_local, uint64_t seed, int64_t offset, hipStream_t st) {
  if (A->use_sc3 && A->real_packed)
    return sc3_random_real(*A->sc3->ly, (double *)x, seed, st, A->sc3->T0, A->sc3->T1, &A->sc3->perm);
  if (A->use_sc3) return sc3_random(*A->sc3->ly, x, seed, st, A->sc3->T0, A->sc3->T1, &A->sc3->perm);
  return vk_random(x, n_local, seed, offset, st, A->right.host.swz);
}

struct ChebFilter {
  int d = 0;
  double c = 0, h = 0, ref = 0;
  void *ta = nullptr, *tb = nullptr;       // two work vectors
  bool on = false;
  double log_tref() const {                // log |T_d((ref - c) / h)|
    const double at = std::fabs((ref - c) / h);
    return at > 1.0 ? d * std::log(at + std::sqrt(at * at - 1.0)) - std::log(2.0) : 0.0;
  }
  // A Ritz pair (mu, absolute residual res_p) of p(A) seen from A: mu = p(lambda) inverted on the wanted side and
  // the residual divided by the slope |p'(lambda)| -- what the residual in A is when the error lies along
  // neighbouring eigenvectors (components deep inside the damped interval count with |mu| / 2h instead; the
  // measured residual decides in the end).  Returns the relative residual estimate, *lam the eigenvalue estimate.
  double seen_from_a(double mu, double res_p, bool low_side, double *lam) const {
    const double a = std::fabs(mu) * std::exp(log_tref());
    if (!(a > 1.0)) { *lam = c; return 1e300; }              // inside the damped interval: not a wanted pair
    const double th = std::acosh(a) / d;
    *lam = low_side ? c - h * std::cosh(th) : c + h * std::cosh(th);
    const double slope = d * std::tanh(d * th) / (h * std::sinh(th)) * std::fabs(mu);
    return res_p / slope / std::max(std::fabs(*lam), 1e-300);
  }
};

struct Ops {
  dnm_mat *A;
  const dnm_hooks *hooks;
  hipStream_t st;
  int64_t n;
  int matvecs = 0;
  ChebFilter *flt = nullptr;
  // real-packed operator (DNM_MAT_REAL_PACKED): the vectors are real, two amplitudes to a complex128 element; the
  // real part of the complex inner product of two such vectors IS their real inner product, its imaginary part
  // means nothing and is dropped wherever an inner product comes back
  bool real = false;

  // y = A x - b z + c2 x, the filter's step (z may be null when b == 0); 2 A x is avoided by halving the
  // recurrence: u_j = s_j / 2^(j-1)  =>  u_{j+1} = (A - c) u_j - (h/2)^2 u_{j-1}, u_1 = (A - c) u_0, u_2 = (A - c) u_1 - (h^2/2) u_0
  int filter_step(const void *x, void *y, const void *z, double b, double c2) {
    if (hooks && hooks->mult) {
      ++matvecs;
      DNM_CHECK(hooks->mult(hooks->ctx, x, y) == 0, "mult hook failed");
      if (b != 0.0) DNM_TRY(vk_axpby(y, z, n, -b, 0.0, 1.0, 0.0, st));
      return vk_axpby(y, x, n, c2, 0.0, 1.0, 0.0, st);
    }
    ++matvecs;
    return dnm_mat_mult_sub2(A, x, y, z ? z : x, b, x, c2, 0.0, (void *)st);
  }
  // y = p(A) x; y must differ from x and from the two work vectors
  // scale_out != null: y is left unscaled and the factor handed back (the caller's next sweep applies it)
  int apply_filter(const void *x, void *y, double *scale_out = nullptr) {
    const ChebFilter &F = *flt;
    // rotate through {ta, tb, y} so that the last term lands in y
    void *buf[3] = {F.ta, F.tb, y};
    const int first = (3 - (F.d % 3)) % 3;              // index of the buffer that takes u_1
    // u_j goes to buf[(first + j - 1) % 3]: u_d -> (first + d - 1) % 3 == 2
    const void *um = nullptr, *uc = x;
    for (int j = 1; j <= F.d; ++j) {
      void *out = buf[(first + j - 1) % 3];
      const double b = j == 1 ? 0.0 : (j == 2 ? 0.5 * F.h * F.h : 0.25 * F.h * F.h);
      DNM_TRY(filter_step(uc, out, um, b, -F.c));
      um = uc;
      uc = out;
    }
    // u_d = h^d T_d / 2^(d-1); normalise by the value at the reference point so that the wanted end is O(1..)
    const double logscale = -(F.d * std::log(F.h) - (F.d - 1) * std::log(2.0)) - F.log_tref();
    if (scale_out) {
      *scale_out = std::exp(logscale);
      return 0;
    }
    return vk_scale(y, n, std::exp(logscale), 0, st);
  }

  int mult(const void *x, void *y) {
    if (flt && flt->on) return apply_filter(x, y);
    ++matvecs;
    if (hooks && hooks->mult) {
      DNM_CHECK(hooks->mult(hooks->ctx, x, y) == 0, "mult hook failed");
      return 0;
    }
    return dnm_mat_mult(A, x, y, (void *)st);
  }
  // y = A x - b z (z may be null), d = <x, y> and, if asked for, nn = |y|^2: the multiply of one three-term
  // Lanczos step
  int mult_dot(const void *x, void *y, zc *d, const void *z = nullptr, double b = 0.0, double *nn = nullptr) {
    double buf[3];
    if (flt && flt->on) {
      // the filter's normalisation rides on the sweep that subtracts b z and takes the sums
      double ys = 1.0;
      DNM_TRY(apply_filter(x, y, &ys));
      DNM_TRY(vec_lanczos_dot_host(y, z, x, n, b, buf, st, ys));
      DNM_TRY(sum(buf, 3));
    } else if (hooks && hooks->mult) {
      DNM_TRY(mult(x, y));
      DNM_TRY(vec_lanczos_dot_host(y, z, x, n, b, buf, st));
      DNM_TRY(sum(buf, 3));
    } else {
      ++matvecs;
      DNM_TRY(dnm_mat_mult_lanczos(A, x, y, z, b, buf, (void *)st));
    }
    *d = zc(buf[0], real ? 0.0 : buf[1]);
    if (nn) *nn = buf[2];
    return 0;
  }
  // y = A x - b z (no inner products)
  int mult_sub(const void *x, void *y, const void *z, double b) {
    if (hooks && hooks->mult) {
      DNM_TRY(mult(x, y));
      return vk_axpby(y, z, n, -b, 0.0, 1.0, 0.0, st);
    }
    ++matvecs;
    return dnm_mat_mult_sub(A, x, y, z, b, (void *)st);
  }
  // y = A x - b z + c z2 (single rank: fused where the kernel allows)
  int mult_sub2(const void *x, void *y, const void *z, double b, const void *z2, zc c) {
    ++matvecs;
    return dnm_mat_mult_sub2(A, x, y, z, b, z2, c.real(), c.imag(), (void *)st);
  }
  int sum(double *buf, int cnt) {
    if (hooks && hooks->allreduce_sum)
      DNM_CHECK(hooks->allreduce_sum(hooks->ctx, buf, cnt) == 0, "allreduce_sum hook failed");
    return 0;
  }
  int maxr(double *buf, int cnt) {
    if (hooks && hooks->allreduce_max)
      DNM_CHECK(hooks->allreduce_max(hooks->ctx, buf, cnt) == 0, "allreduce_max hook failed");
    return 0;
  }
  // h = V[:, 0:nv)^H w (global)
  int mdot(const void *V, int nv, const void *w, std::vector<zc> &h) {
    std::vector<double> buf((size_t)2 * nv);
    DNM_TRY(vec_mdot_host(V, n, nv, w, n, buf.data(), st));
    DNM_TRY(sum(buf.data(), 2 * nv));
    h.resize(nv);
    for (int j = 0; j < nv; ++j) h[j] = zc(buf[2 * j], real ? 0.0 : buf[2 * j + 1]);
    return 0;
  }
  int norm(const void *w, double *out) {
    double buf[2];
    DNM_TRY(vec_mdot_host(w, n, 1, w, n, buf, st));
    DNM_TRY(sum(buf, 1));
    *out = std::sqrt(buf[0] > 0 ? buf[0] : 0.0);
    return 0;
  }
  // w += V[:, 0:nv) c
  int maxpy(void *w, const void *V, int nv, const std::vector<zc> &c) {
    std::vector<double> buf((size_t)2 * nv);
    for (int j = 0; j < nv; ++j) { buf[2 * j] = c[j].real(); buf[2 * j + 1] = c[j].imag(); }
    const double *cd = nullptr;
    DNM_TRY(vec_upload_coefs(buf.data(), buf.size(), st, &cd));
    return vk_maxpy(w, V, n, nv, n, cd, st);
  }
  // orthogonalise p against V[:, 0:nv): classical Gram-Schmidt with one
  // refinement pass when needed (the DGKS test SLEPc's BV uses by default,
  // eta = 1/sqrt(2)); coefficients accumulate in h; returns ||p|| afterwards
  int orthogonalize(void *p, const void *V, int nv, std::vector<zc> &h, double *nrm, int min_passes = 1) {
    std::vector<zc> h1, neg;
    h.assign(nv, zc(0));
    if (nv > 4) {
      // H is Hermitian: p = A v_j is dominated by its components along the last
      // two basis vectors.  Removing those first (two vectors, cheap) leaves a
      // full pass that rarely needs refinement.
      const int lo = nv - 2;
      const void *Vl = (const char *)V + (size_t)lo * (size_t)n * 16;
      DNM_TRY(mdot(Vl, 2, p, h1));
      neg.assign(2, zc(0));
      for (int j = 0; j < 2; ++j) { neg[j] = -h1[j]; h[lo + j] += h1[j]; }
      DNM_TRY(maxpy(p, Vl, 2, neg));
    }
    for (int pass = 0; pass < 3; ++pass) {
      DNM_TRY(mdot(V, nv, p, h1));
      neg.resize(nv);
      double hn2 = 0.0;
      for (int j = 0; j < nv; ++j) { neg[j] = -h1[j]; h[j] += h1[j]; hn2 += std::norm(h1[j]); }
      DNM_TRY(maxpy(p, V, nv, neg));
      DNM_TRY(norm(p, nrm));
      const double before = std::sqrt((*nrm) * (*nrm) + hn2);   // ||p|| before this pass
      if (pass + 1 >= min_passes && *nrm >= 0.7071067811865476 * before) break;   // no cancellation: done
    }
    return 0;
  }
  // First vector of a thick-restart cycle, p = A q_l against [u_0..u_{l-1}, q_l]: the components along the kept
  // Ritz vectors are the spikes s_i of the projected matrix (A u_i = theta_i u_i + s_i q_l), so the first
  // Gram-Schmidt pass needs no inner products except <q_l, p>; a measured pass follows (the Ritz vectors of a
  // semi-orthogonal basis are orthonormal to sqrt(eps) only) and its coefficients correct h.
  int orthogonalize_known(void *p, const void *V, int nv, const std::vector<double> &spike, std::vector<zc> &h,
                          double *nrm) {
    std::vector<zc> h1, neg(nv);
    h.assign(nv, zc(0));
    const void *ql = (const char *)V + (size_t)(nv - 1) * (size_t)n * 16;
    DNM_TRY(mdot(ql, 1, p, h1));
    for (int i = 0; i + 1 < nv; ++i) h[i] = zc(spike[(size_t)i], 0.0);
    h[nv - 1] = h1[0];
    for (int i = 0; i < nv; ++i) neg[i] = -h[i];
    DNM_TRY(maxpy(p, V, nv, neg));
    for (int pass = 0; pass < 2; ++pass) {
      DNM_TRY(mdot(V, nv, p, h1));
      double hn2 = 0.0;
      for (int j = 0; j < nv; ++j) { neg[j] = -h1[j]; h[j] += h1[j]; hn2 += std::norm(h1[j]); }
      DNM_TRY(maxpy(p, V, nv, neg));
      DNM_TRY(norm(p, nrm));
      const double before = std::sqrt((*nrm) * (*nrm) + hn2);
      if (*nrm >= 0.7071067811865476 * before) break;
    }
    return 0;
  }
};

// Simon's omega-recurrence: a running estimate of |v_{j+1}^H v_k| for a Lanczos
// process without re-orthogonalisation.  While every estimate stays below
// sqrt(eps) the three-term recurrence is kept (5 vector passes per step);
// when one crosses it the new vector and its successor are orthogonalised
// against the whole basis (partial re-orthogonalisation, Simon 1984).
// Components removed by a re-orthogonalisation pass are not recorded in the projected
// matrix, so a Ritz pair's true residual exceeds its estimate by about
// (level at which the pass is triggered) x |H|.  The trigger level therefore follows the
// requested tolerance: tol/10, at most sqrt(eps) (Simon's semi-orthogonality bound), at
// least a few times the rounding floor eps1 of a dot product (below that every step is a
// full pass, which is what full re-orthogonalisation achieves anyway).
static double pro_threshold(double eps1, double tol) {
  double t = std::sqrt(2.220446049250313e-16);
  if (tol > 0 && 0.1 * tol < t) t = 0.1 * tol;
  if (t < 4.0 * eps1) t = 4.0 * eps1;
  if (knob("DNM_PRO_THRESH")) t = atof(knob("DNM_PRO_THRESH"));
  return t;
}

struct LanczosMonitor {
  std::vector<double> alpha, beta;      // alpha[j]; beta[j] = ||r_{j-1}|| (beta[0] = 0)
  std::vector<double> wprev, wcur;      // omega_{j-1,.}, omega_{j,.}
  double eps1 = 0, thresh = 0;
  bool force_next = false;
  int reorths = 0;
  void reset(int m, double n_global, double tol) {
    alpha.assign(m + 2, 0.0);
    beta.assign(m + 2, 0.0);
    wprev.assign(m + 2, 0.0);
    wcur.assign(m + 2, 0.0);
    wcur[0] = 1.0;
    const double eps = 2.220446049250313e-16;
    eps1 = eps * std::sqrt(n_global) / 2.0;
    if (eps1 > 1e-11) eps1 = 1e-11;
    thresh = pro_threshold(eps1, tol);
    force_next = false;
  }
  // step j produced alpha_j and beta_{j+1}; returns true when v_{j+1} needs a full pass
  bool update(int j, double a_j, double b_next) {
    alpha[j] = a_j;
    beta[j + 1] = b_next;
    std::vector<double> wnew(wcur.size(), 0.0);
    double worst = 0.0;
    if (b_next > 0) {
      for (int k = 0; k < j; ++k) {
        double v = beta[k + 1] * wcur[k + 1] + (alpha[k] - a_j) * wcur[k] - beta[j] * wprev[k];
        if (k > 0) v += beta[k] * wcur[k - 1];
        v = (v + (v >= 0 ? eps1 : -eps1)) / b_next;
        wnew[k] = v;
        worst = std::max(worst, std::fabs(v));
      }
    }
    if (j >= 0) wnew[j] = eps1;
    wnew[j + 1] = 1.0;
    wprev.swap(wcur);
    wcur.swap(wnew);
    const bool need = force_next || worst > thresh;
    if (need) {
      force_next = !force_next;          // the successor of a re-orthogonalised vector gets a pass too
      for (int k = 0; k <= j; ++k) wcur[k] = eps1;
      ++reorths;
    }
    return need;
  }
};

// The same estimate for thick-restart Lanczos.  The basis of a cycle is
// q_0..q_{l-1} (kept Ritz vectors, H u_i = theta_i u_i + s_i q_l), then Lanczos vectors
// q_l, q_{l+1}, ...; with T the projected matrix (diag(theta) + spike row/column l +
// tridiagonal beyond), omega_{j+1,i} = q_{j+1}^H q_i obeys
//   beta_{j+1} omega_{j+1,i} = sum_k T_{k,i} omega_{j,k} - alpha_j omega_{j,i} - beta_j omega_{j-1,i}   (j > l),
// the step j = l being orthogonalised against the whole basis explicitly (it has to
// remove the spike components anyway).
struct RestartMonitor {
  int l = 0;
  std::vector<double> th, sp, alpha, beta, wprev, wcur;
  double eps1 = 0, thresh = 0;
  bool force_next = false;
  int reorths = 0, steps = 0;
  void init(int m, double n_global, double tol) {
    const double eps = 2.220446049250313e-16;
    eps1 = eps * std::sqrt(n_global) / 2.0;
    if (eps1 > 1e-11) eps1 = 1e-11;
    thresh = pro_threshold(eps1, tol);
    alpha.assign(m + 2, 0.0);
    beta.assign(m + 2, 0.0);
    wprev.assign(m + 2, 0.0);
    wcur.assign(m + 2, 0.0);
  }
  // start of a cycle: row_l[i] bounds |q_l^H u_i|
  void begin_cycle(int l_, const std::vector<double> &theta, const std::vector<double> &spike,
                   const std::vector<double> &row_l) {
    l = l_;
    th = theta;
    sp = spike;
    std::fill(wprev.begin(), wprev.end(), 0.0);
    std::fill(wcur.begin(), wcur.end(), 0.0);
    for (int i = 0; i < l; ++i) wcur[i] = std::max(eps1, i < (int)row_l.size() ? row_l[i] : eps1);
    wcur[l] = 1.0;
    force_next = false;
  }
  // step j = l was orthogonalised against q_0..q_l explicitly
  void first_step(double a_l, double b_next) {
    alpha[l] = a_l;
    beta[l + 1] = b_next;
    wprev = wcur;
    std::fill(wcur.begin(), wcur.end(), 0.0);
    for (int i = 0; i <= l; ++i) wcur[i] = eps1;
    wcur[l + 1] = 1.0;
  }
  // step j > l produced alpha_j, beta_{j+1} by the three-term recurrence; true: q_{j+1} needs a full pass
  bool update(int j, double a_j, double b_next) {
    alpha[j] = a_j;
    beta[j + 1] = b_next;
    ++steps;
    std::vector<double> wnew(wcur.size(), 0.0);
    double worst = 0.0;
    if (b_next > 0) {
      for (int i = 0; i < j; ++i) {
        double v;
        if (i < l) {
          v = th[i] * wcur[i] + sp[i] * wcur[l];
        } else if (i == l) {
          v = alpha[l] * wcur[l] + beta[l + 1] * wcur[l + 1];
          for (int k = 0; k < l; ++k) v += sp[k] * wcur[k];
        } else {
          v = beta[i] * wcur[i - 1] + alpha[i] * wcur[i] + beta[i + 1] * wcur[i + 1];
        }
        v -= a_j * wcur[i] + beta[j] * wprev[i];
        v = (v + (v >= 0 ? eps1 : -eps1)) / b_next;
        wnew[i] = v;
        worst = std::max(worst, std::fabs(v));
      }
    }
    wnew[j] = eps1;
    wnew[j + 1] = 1.0;
    wprev.swap(wcur);
    wcur.swap(wnew);
    const bool need = force_next || worst > thresh;
    if (need) {
      force_next = !force_next;
      for (int k = 0; k <= j; ++k) wcur[k] = eps1;
      ++reorths;
    }
    return need;
  }
};

// Krylov basis workspace, kept between solves (hipMalloc/hipFree of tens of GiB per
// call costs more than a solve); dnm_release_workspace() returns it.
static DevBuf g_basis;
static int basis_workspace(size_t bytes, void **p) {
  if (g_basis.bytes < bytes) {
    g_basis.release();
    DNM_TRY(g_basis.alloc(bytes));
  }
  *p = g_basis.p;
  return 0;
}

static char *vecptr(void *base, int64_t n, int j) { return (char *)base + (size_t)j * (size_t)n * 16; }

static double round2(double t) {
  // Expokit's two-significant-digit rounding of a step size
  const double sqr1 = std::sqrt(0.1);
  const double p1 = std::pow(10.0, std::round(std::log10(t) - sqr1) - 1.0);
  return std::trunc(t / p1 + 0.55) * p1;
}

}  // namespace dnm

using namespace dnm;

extern "C" {

int dnm_workspace_bytes(size_t *bytes) {
  DNM_CHECK(bytes, "null argument");
  *bytes = g_basis.bytes;
  return 0;
}

int dnm_workspace_reserve(size_t bytes, void *stream) {
  if (g_basis.bytes >= bytes) return 0;
  void *p = nullptr;
  DNM_TRY(basis_workspace(bytes, &p));
  DNM_HIP(hipMemsetAsync(p, 0, bytes, (hipStream_t)stream));
  DNM_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int dnm_release_workspace(void) {
  g_basis.release();
  rdm_release_scratch();
  return 0;
}

}  // extern "C"

// Chebyshev coefficients of one step: J_k(z) for k = 0..K with the tail beyond K below `cut`
static int cheb_coeffs(double z, double cut, std::vector<double> &J, double *tail_out) {
  const int kmax = (int)(z + 30.0 * std::cbrt(z + 1.0) + 80.0);
  J.resize((size_t)kmax + 1);
  for (int k = 0; k <= kmax; ++k) J[(size_t)k] = std::cyl_bessel_j((double)k, z);
  double tail = 0;
  int K = kmax;
  while (K > 1 && tail + 2.0 * std::fabs(J[(size_t)K]) < cut) { tail += 2.0 * std::fabs(J[(size_t)K]); --K; }
  DNM_CHECK(K < kmax, "internal: Bessel coefficients have not decayed (z = %g)", z);
  J.resize((size_t)K + 1);
  if (tail_out) *tail_out = tail;
  return 0;
}

// steps of |r t| <= 64: the coefficients J_k(z) die out super-exponentially beyond k = z
static void cheb_steps(double ztot, int *nsteps, double *z) {
  *nsteps = std::max(1, (int)std::ceil(ztot / 64.0));
  *z = ztot / *nsteps;
}

// multiplies the expansion needs for |r t| = ztot at tolerance tol
static int cheb_cost(double ztot, double tol, int64_t *terms) {
  int nsteps;
  double z;
  cheb_steps(ztot, &nsteps, &z);
  std::vector<double> J;
  DNM_TRY(cheb_coeffs(z, tol / (100.0 * nsteps), J, nullptr));
  *terms = (int64_t)nsteps * (int64_t)(J.size() - 1);
  return 0;
}

// y <- exp(-i t A) y by the Chebyshev expansion; r >= spectral radius; W: four work vectors
static int cheb_core(Ops &ops, void *y, int64_t n_local, double t, double tol, double r, void *W, int *steps_out,
                     double *err_out) {
  hipStream_t st = ops.st;
  int nsteps;
  double z;
  cheb_steps(std::fabs(r * t), &nsteps, &z);
  std::vector<double> J;
  double tail = 0;
  DNM_TRY(cheb_coeffs(z, tol / (100.0 * nsteps), J, &tail));
  const int K = (int)J.size() - 1;
  // a_k = (2 - delta_k0) (-i sgn t)^k J_k(z)
  const zc mi(0.0, t > 0 ? -1.0 : 1.0);
  auto coef = [&](int k) {
    const zc pw = (k & 3) == 0 ? zc(1, 0) : (k & 3) == 1 ? mi : (k & 3) == 2 ? zc(-1, 0) : -mi;
    return (k == 0 ? 1.0 : 2.0) * J[(size_t)k] * pw;
  };
  if (steps_out) *steps_out = nsteps;
  if (err_out) *err_out = tail * nsteps;

  const char *cenv = knob("DNM_CHEB_FORM");
  const bool clenshaw = !(ops.hooks && ops.hooks->mult) && dnm_mat_fuses_init(ops.A) && !(cenv && cenv[0] == 'f');
  if (clenshaw) {
    // Clenshaw's backward recurrence, b_k = a_k x + (2/r) A b_{k+1} - b_{k+2}, result a_0 x + A b_1 / r - b_2:
    // every term is ONE fused multiply (the a_k x term and the b_{k+2} term start the accumulators), three
    // work vectors, x is the state itself.  Unnormalised: B_k = b_k / gamma_k, gamma_K = 1, gamma_k = (2/r) gamma_{k+1}:
    //   B_k = A B_{k+1} - (r/2)^2 B_{k+2} + (a_k / gamma_k) x
    const double beta2 = 0.25 * r * r;
    auto sl = [&](int k) { return (void *)vecptr(W, n_local, k % 3); };
    for (int step = 0; step < nsteps; ++step) {
      double g[3];
      const zc aK = coef(K);
      DNM_TRY(vk_axpby(sl(K), y, n_local, aK.real(), aK.imag(), 0.0, 0.0, st));       // B_K = a_K x
      g[K % 3] = 1.0;
      for (int k = K - 1; k >= 1; --k) {
        const double gk = g[(k + 1) % 3] * (2.0 / r);
        const zc c = coef(k) / gk;
        if (k == K - 1) DNM_TRY(ops.mult_sub2(sl(k + 1), sl(k), y, 0.0, y, c));          // b_{K+1} = 0
        else DNM_TRY(ops.mult_sub2(sl(k + 1), sl(k), sl(k + 2), beta2, y, c));
        g[k % 3] = gk;
        if (gk < 1e-200 || gk > 1e200) {      // (tiny or huge norm bound r) bring the two live vectors back to O(1) by the same factor
          DNM_TRY(vk_scale(sl(k), n_local, gk, 0.0, st));
          DNM_TRY(vk_scale(sl(k + 1), n_local, gk, 0.0, st));
          g[(k + 1) % 3] /= gk;
          g[k % 3] = 1.0;
        }
      }
      // result = a_0 x + (gamma_1 / r) [A B_1 - (r^2/2) B_2]  (gamma_2 / gamma_1 = r / 2)
      const double g1 = g[1 % 3];
      const zc c0 = coef(0) * (r / g1);
      if (K >= 2) DNM_TRY(ops.mult_sub2(sl(1), sl(0), sl(2), 0.5 * r * r, y, c0));
      else DNM_TRY(ops.mult_sub2(sl(1), sl(0), y, 0.0, y, c0));
      DNM_TRY(vk_axpby(y, sl(0), n_local, g1 / r, 0.0, 0.0, 0.0, st));
    }
    return 0;
  }
  // ring of four vectors U_k = T_k(A/r) x / gamma_k in slot k & 3, gamma_0 = 1, gamma_{k+1} = (2/r) gamma_k:
  //   T_{k+1} = (2/r) A T_k - T_{k-1}   <=>   U_{k+1} = A U_k - (r/2)^2 U_{k-1}
  const double beta = 0.25 * r * r;
  auto slot = [&](int k) { return (void *)vecptr(W, n_local, k & 3); };
  for (int step = 0; step < nsteps; ++step) {
    double gam[4];
    // U_0 = the state, U_1 = A U_0 / 2 (T_1 = A x / r = (2/r) U_1)
    DNM_TRY(vk_copy(slot(0), y, n_local, st));
    DNM_TRY(ops.mult(slot(0), slot(1)));
    DNM_TRY(vk_scale(slot(1), n_local, 0.5, 0.0, st));
    gam[0] = 1.0;
    gam[1] = 2.0 / r;
    {   // y holds U_0 already: y += (a_0 - 1) U_0 + a_1 gamma_1 U_1
      std::vector<zc> c = {coef(0) - 1.0, coef(1) * gam[1]};
      DNM_TRY(ops.maxpy(y, slot(0), 2, c));
    }
    for (int k = 1; k < K; ++k) {
      DNM_TRY(ops.mult_sub(slot(k), slot(k + 1), slot(k - 1), beta));
      gam[(k + 1) & 3] = gam[k & 3] * (2.0 / r);
      if (gam[(k + 1) & 3] < 1e-200 || gam[(k + 1) & 3] > 1e200) {
        // bring the two live vectors back to O(1) by the same factor: the recurrence is unchanged
        const double f = gam[(k + 1) & 3];
        DNM_TRY(vk_scale(slot(k + 1), n_local, f, 0.0, st));
        DNM_TRY(vk_scale(slot(k), n_local, f, 0.0, st));
        gam[(k + 1) & 3] /= f;
        gam[k & 3] /= f;
      }
      // terms k (even) and k + 1 sit side by side in the ring: one sweep adds both; an even K leaves its last
      // term alone
      if ((k + 1) & 1) {
        std::vector<zc> c = {coef(k) * gam[k & 3], coef(k + 1) * gam[(k + 1) & 3]};
        DNM_TRY(ops.maxpy(y, slot(k), 2, c));
      } else if (k + 1 == K) {
        std::vector<zc> c = {coef(K) * gam[K & 3]};
        DNM_TRY(ops.maxpy(y, slot(K), 1, c));
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------
// Basis-free Lanczos for ONE extremal eigenpair (memory-bound sizes)
// ---------------------------------------------------------------------------
// A symmetric tridiagonal matrix T (diagonal a[0..n), off-diagonal b[0..n-1)): number of eigenvalues below x
static int sturm_count(const std::vector<double> &a, const std::vector<double> &b, int n, double x) {
  int cnt = 0;
  double q = a[0] - x;
  if (q < 0) ++cnt;
  for (int i = 1; i < n; ++i) {
    const double den = std::fabs(q) > 1e-300 ? q : (q < 0 ? -1e-300 : 1e-300);
    q = a[i] - x - b[i - 1] * b[i - 1] / den;
    if (q < 0) ++cnt;
  }
  return cnt;
}

// k-th smallest eigenvalue of T by bisection, its eigenvector (unit norm) by inverse iteration
static double tridiag_eigpair(const std::vector<double> &a, const std::vector<double> &b, int n, int k,
                              std::vector<double> &z) {
  double lo = a[0], hi = a[0], nrm = 0;
  for (int i = 0; i < n; ++i) {
    const double r = (i > 0 ? std::fabs(b[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(b[i]) : 0.0);
    lo = std::min(lo, a[i] - r);
    hi = std::max(hi, a[i] + r);
    nrm = std::max(nrm, std::fabs(a[i]) + r);
  }
  for (int it = 0; it < 200 && hi - lo > 4e-16 * std::max(nrm, 1e-300); ++it) {
    const double mid = 0.5 * (lo + hi);
    if (sturm_count(a, b, n, mid) > k) hi = mid; else lo = mid;
  }
  const double theta = 0.5 * (lo + hi);
  z.assign(n, 0.0);
  if (n == 1) { z[0] = 1.0; return theta; }
  // (T - theta') z = rhs by Gaussian elimination with partial pivoting on the tridiagonal (theta' a hair off the
  // eigenvalue); three sweeps from a generic start
  const double shift = theta + 1e-13 * std::max(nrm, 1e-300);
  std::vector<double> rhs(n);
  for (int i = 0; i < n; ++i) rhs[i] = 1.0 / std::sqrt((double)n) * ((i & 1) ? 0.7 : 1.0);
  std::vector<double> d(n), du(n), du2(n), dl(n);
  for (int sweep = 0; sweep < 3; ++sweep) {
    for (int i = 0; i < n; ++i) { d[i] = a[i] - shift; du[i] = i + 1 < n ? b[i] : 0.0; dl[i] = i + 1 < n ? b[i] : 0.0; du2[i] = 0.0; }
    z = rhs;
    for (int i = 0; i + 1 < n; ++i) {
      if (std::fabs(dl[i]) > std::fabs(d[i])) {          // swap rows i and i+1
        std::swap(d[i], dl[i]);
        const double t = du[i]; du[i] = d[i + 1]; d[i + 1] = t;
        du2[i] = du[i + 1]; du[i + 1] = 0.0;
        std::swap(z[i], z[i + 1]);
        // after the swap: row i = (d[i], du[i], du2[i]), row i+1 = (dl[i], d[i+1], du[i+1])
      }
      const double piv = std::fabs(d[i]) > 1e-300 ? d[i] : 1e-300;
      const double f = dl[i] / piv;
      d[i + 1] -= f * du[i];
      du[i + 1] -= f * du2[i];
      z[i + 1] -= f * z[i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double v = z[i];
      if (i + 1 < n) v -= du[i] * z[i + 1];
      if (i + 2 < n) v -= du2[i] * z[i + 2];
      const double piv = std::fabs(d[i]) > 1e-300 ? d[i] : 1e-300;
      z[i] = v / piv;
    }
    double nn = 0;
    for (int i = 0; i < n; ++i) nn += z[i] * z[i];
    nn = std::sqrt(nn);
    for (int i = 0; i < n; ++i) z[i] /= nn;
    rhs = z;
  }
  return theta;
}

// Lanczos without a stored basis: three work vectors, the three-term recurrence only, the tridiagonal matrix on
// the host.  The extremal Ritz value converges regardless of the loss of orthogonality (Paige); copies of it that
// appear later do not matter because the iteration stops at convergence.  The Ritz vector, if wanted, is built
// in a second run of the same recurrence with the recorded coefficients.  At 16 GiB per vector a step is the
// multiply plus one sweep -- no restarts, no re-orthogonalisation passes over a 240 GiB basis.
//
// nev > 1 (problems whose vectors leave no room for a restarted basis: the 36-site kagome torus, 34 GiB per real
// vector): one pair after the other, each by the same recurrence on the operator deflated by the pairs found --
// every new Lanczos vector is projected against them (one fused inner-product sweep + one update sweep per step).
// A found vector is an eigenvector to the residual tol, so the deflated operator's extremal pair is the next pair
// of H to second order in tol; unlike one Krylov space, deflation also returns every copy of a degenerate level.
// Memory: 4 work vectors + the nev vectors (in `evecs` when the caller wants them, else nev - 1 in the workspace).
static int eigsolve_basis_free(Ops &ops, dnm_mat *A, int64_t n_local, int nev, int which, double tol, int max_steps,
                               uint64_t seed, const dnm_hooks *hooks, double *evals, void *evecs,
                               dnm_solver_stats *stats, hipStream_t st) {
  void *W = nullptr;
  DNM_TRY(basis_workspace((size_t)(4 + (evecs ? 0 : nev - 1)) * (size_t)n_local * 16, &W));
  const int64_t offset = hooks ? A->row0 : 0;
  void *F = evecs ? evecs : (void *)vecptr(W, n_local, 4);        // the pairs found so far, contiguous
  const char *venv = knob("DNM_EIGS_VERIFY");
  const bool verify = venv && venv[0] == '1';
  int nconv = 0, total_steps = 0;
  double worst = 0;
  for (int e = 0; e < nev; ++e) {
    // where this pair's vector goes (the last one is not needed unless the caller wants it)
    void *dst = evecs ? (void *)vecptr(evecs, n_local, e) : (e + 1 < nev ? (void *)vecptr(W, n_local, 4 + e) : nullptr);
    std::vector<double> al, be, svec;
    struct Step { double are, aim, s1, s2; };      // what the update of a step did, so that the second run repeats it
    std::vector<Step> rec;
    std::vector<zc> proj;                          // [step][found]: the projections taken out, for the second run
    double theta = 0, res = 0, err = 0;
    bool converged = false;
    int steps = 0, rounds = 0;
    auto slot = [&](int k) { return (void *)vecptr(W, n_local, k % 3); };
    // p <- p - F F^H p (first run: coefficients measured and recorded; second run: the recorded ones), |p|^2 after.
    // At EVERY step: projecting only every fourth one (a quarter of the sweeps) lets the recurrence alternate between H
    // and the deflated operator, and once the first Ritz value has converged the tridiagonal matrix is that of no
    // symmetric operator any more -- Ritz values far below the spectrum (measured: docs/lab/r05.md section 10).
    auto deflate = [&](void *p, int j, bool replay, double *n2) -> int {
      std::vector<zc> h;
      const size_t at = (size_t)j * e;
      if (replay) h.assign(proj.begin() + at, proj.begin() + at + e);
      else {
        DNM_TRY(ops.mdot(F, e, p, h));
        proj.insert(proj.end(), h.begin(), h.end());
      }
      if (e > 1) {
        std::vector<zc> neg(e - 1);
        for (int i = 0; i + 1 < e; ++i) neg[i] = -h[i];
        DNM_TRY(ops.maxpy(p, F, e - 1, neg));
      }
      DNM_TRY(vec_lanczos_update_host(p, vecptr(F, n_local, e - 1), nullptr, n_local, h[e - 1].real(), h[e - 1].imag(),
                                      0.0, n2, st, 1.0));
      return ops.sum(n2, 1);
    };
    // start vector: seeded normal deviates, or (later rounds) the Ritz vector of the round before, kept in `dst`;
    // without its components along the pairs found
    auto start = [&](bool from_prev) -> int {
      if (from_prev) DNM_TRY(vk_axpby(slot(0), dst, n_local, 1.0, 0.0, 0.0, 0.0, st));
      else DNM_TRY(random_start(A, slot(0), n_local, seed + 7919u * (uint64_t)e, offset, st));
      for (int pass = 0; pass < (e > 0 ? 2 : 0); ++pass) {
        std::vector<zc> h, neg(e);
        DNM_TRY(ops.mdot(F, e, slot(0), h));
        for (int i = 0; i < e; ++i) neg[i] = -h[i];
        DNM_TRY(ops.maxpy(slot(0), F, e, neg));
      }
      double n0 = 0;
      DNM_TRY(ops.norm(slot(0), &n0));
      DNM_CHECK(n0 > 0, "zero start vector");
      return vk_scale(slot(0), n_local, 1.0 / n0, 0, st);
    };
    auto pick = [&](int n, std::vector<double> &z) {
      if (which == DNM_WHICH_LOWEST) return tridiag_eigpair(al, be, n, 0, z);
      if (which == DNM_WHICH_HIGHEST) return tridiag_eigpair(al, be, n, n - 1, z);
      std::vector<double> z2;
      const double lo = tridiag_eigpair(al, be, n, 0, z), hi = tridiag_eigpair(al, be, n, n - 1, z2);
      if (std::fabs(hi) > std::fabs(lo)) { z = z2; return hi; }
      return lo;
    };
    bool measured = false;
    while (true) {
      al.clear(); be.clear(); rec.clear(); proj.clear();
      converged = false;
      steps = 0;
      DNM_TRY(start(rounds > 0));
      for (int j = 0; j < max_steps; ++j) {
        void *q = slot(j), *p = slot(j + 1), *qm = slot(j + 2);     // (j + 2) % 3 == (j - 1) % 3
        zc d(0);
        double pn2 = 0;
        DNM_TRY(ops.mult_dot(q, p, &d, j > 0 ? qm : nullptr, j > 0 ? be[j - 1] : 0.0, &pn2));
        al.push_back(d.real());
        const double b2 = pn2 - std::norm(d);
        const bool fused = b2 > 1e-4 * pn2 && pn2 > 0;
        double n2 = 0, bn;
        Step sr{d.real(), d.imag(), fused ? 1.0 / std::sqrt(b2) : 1.0, 0.0};
        DNM_TRY(vec_lanczos_update_host(p, q, nullptr, n_local, sr.are, sr.aim, 0.0, &n2, st, sr.s1));
        DNM_TRY(ops.sum(&n2, 1));
        if (e > 0) DNM_TRY(deflate(p, j, false, &n2));
        if (fused) {
          const double nu = std::sqrt(n2 > 0 ? n2 : 0.0);
          bn = std::sqrt(b2) * nu;
          if (std::fabs(n2 - 1.0) > 1e-12 && nu > 0) sr.s2 = 1.0 / nu;
        } else {
          bn = std::sqrt(n2 > 0 ? n2 : 0.0);
          if (bn > 0) sr.s2 = 1.0 / bn;
        }
        if (sr.s2 != 0.0) DNM_TRY(vk_scale(p, n_local, sr.s2, 0, st));
        rec.push_back(sr);
        be.push_back(bn);
        steps = j + 1;
        double scale = 0;
        for (int i = 0; i < steps; ++i) scale = std::max(scale, std::fabs(al[i]) + be[i]);
        const bool breakdown = bn <= 1e-14 * std::max(1.0, scale);
        if (steps >= 8 || breakdown || steps == max_steps) {
          theta = pick(steps, svec);
          res = std::fabs(bn * svec[steps - 1]);
          if (breakdown || res <= tol * std::max(std::fabs(theta), 1e-300)) { converged = true; break; }
        }
      }
      total_steps += steps;
      evals[e] = theta;
      err = res / std::max(std::fabs(theta), 1e-300);
      if (!(dst || verify) || steps == 0) break;
      // second run: the same vectors from the same start by the same arithmetic (recorded coefficients and scales),
      // v = sum_j s_j q_j accumulated in the fourth slot
      void *v = vecptr(W, n_local, 3);
      DNM_TRY(start(rounds > 0));
      DNM_TRY(vk_axpby(v, slot(0), n_local, svec[0], 0.0, 0.0, 0.0, st));
      for (int j = 0; j + 1 < steps; ++j) {
        void *q = slot(j), *p = slot(j + 1), *qm = slot(j + 2);
        if (j > 0) DNM_TRY(ops.mult_sub(q, p, qm, be[j - 1]));
        else DNM_TRY(ops.mult(q, p));
        double n2 = 0;
        DNM_TRY(vec_lanczos_update_host(p, q, nullptr, n_local, rec[j].are, rec[j].aim, 0.0, &n2, st, rec[j].s1));
        if (e > 0) DNM_TRY(deflate(p, j, true, &n2));
        if (rec[j].s2 != 0.0) DNM_TRY(vk_scale(p, n_local, rec[j].s2, 0, st));
        DNM_TRY(vk_axpby(v, p, n_local, svec[j + 1], 0.0, 1.0, 0.0, st));
      }
      double vn = 0;
      DNM_TRY(ops.norm(v, &vn));
      DNM_CHECK(vn > 0, "zero Ritz vector");
      DNM_TRY(vk_scale(v, n_local, 1.0 / vn, 0, st));
      // what was promised, measured on H itself (not the deflated operator): |H v - <v, H v> v| / |theta|
      zc d(0);
      void *hv = slot(0);
      DNM_TRY(ops.mult_dot(v, hv, &d));
      double n2 = 0;
      DNM_TRY(vec_lanczos_update_host(hv, v, nullptr, n_local, d.real(), d.imag(), 0.0, &n2, st));
      DNM_TRY(ops.sum(&n2, 1));
      evals[e] = d.real();
      err = std::sqrt(n2 > 0 ? n2 : 0.0) / std::max(std::fabs(evals[e]), 1e-300);
      measured = true;
      if (dst) DNM_TRY(vk_copy(dst, v, n_local, st));
      // the contract is a residual below tol (computations.py:274-275 raises otherwise): the estimate of the first
      // run is not the vector's residual once rounding has crept into a long recurrence.  Polish: Lanczos again from
      // the Ritz vector itself (a handful of steps); a vector that still misses tol is reported as not converged.
      if (!converged || !dst || err <= tol) break;
      if (++rounds >= 3) { converged = false; break; }
    }
    if (knob("DNM_KRYLOV_DEBUG"))
      fprintf(stderr, "dnm_eigsolve (basis-free Lanczos, pair %d of %d): %d steps, %d matvecs in all, theta = %.12g, relative residual %.2e (%s)\n",
              e + 1, nev, steps, ops.matvecs, evals[e], err, measured ? "measured" : "Lanczos estimate");
    worst = std::max(worst, err);
    if (!converged) break;
    ++nconv;
  }
  DNM_HIP(hipStreamSynchronize(st));
  stats->its = nev;
  stats->matvecs = ops.matvecs;
  stats->err_est = worst;
  stats->nconv = nconv;
  stats->reason = nconv == nev ? DNM_CONVERGED_TOL : DNM_DIVERGED_ITS;
  (void)total_steps;
  return 0;
}

// Spectral extent of A as the vector x sees it: k Lanczos steps from x (three work vectors in W, x untouched), the
// larger magnitude of the extreme Ritz values -- a lower bound of the spectral radius that is close after a few
// steps.  0 if the recurrence breaks down (x lies in a small invariant subspace: a Krylov method is exact there).
static int lanczos_extent(Ops &ops, const void *x, double xnorm, int64_t n_local, void *W, int k, double *rho,
                          hipStream_t st) {
  std::vector<double> al, be, z;
  auto slot = [&](int j) { return (void *)vecptr(W, n_local, j % 3); };
  DNM_TRY(vk_axpby(slot(0), x, n_local, 1.0 / xnorm, 0.0, 0.0, 0.0, st));
  *rho = 0.0;
  for (int j = 0; j < k; ++j) {
    void *q = slot(j), *p = slot(j + 1), *qm = slot(j + 2);
    zc d(0);
    double pn2 = 0;
    DNM_TRY(ops.mult_dot(q, p, &d, j > 0 ? qm : nullptr, j > 0 ? be[j - 1] : 0.0, &pn2));
    al.push_back(d.real());
    double n2 = 0;
    DNM_TRY(vec_lanczos_update_host(p, q, nullptr, n_local, d.real(), d.imag(), 0.0, &n2, st, 1.0));
    DNM_TRY(ops.sum(&n2, 1));
    const double bn = std::sqrt(n2 > 0 ? n2 : 0.0);
    double scale = 0;
    for (size_t i = 0; i < al.size(); ++i) scale = std::max(scale, std::fabs(al[i]) + (i < be.size() ? be[i] : 0.0));
    if (bn <= 1e-10 * std::max(1.0, scale)) return 0;
    DNM_TRY(vk_scale(p, n_local, 1.0 / bn, 0, st));
    be.push_back(bn);
  }
  const int n = (int)al.size();
  const double lo = tridiag_eigpair(al, be, n, 0, z), hi = tridiag_eigpair(al, be, n, n - 1, z);
  *rho = std::max(std::fabs(lo), std::fabs(hi));
  return 0;
}

extern "C" {

int dnm_expm_chebyshev(dnm_mat *A, const void *x, void *y, int64_t n_local, double t, double tol,
                       const dnm_hooks *hooks, dnm_solver_stats *stats, void *stream) {
  DNM_CHECK(A && x && y && stats, "null argument");
  DNM_CHECK(!A->real_packed, "exp(-iHt) needs complex vectors: not for a real-packed operator");
  hipStream_t st = (hipStream_t)stream;
  Ops ops{A, hooks, st, n_local};
  stats->reason = 0; stats->its = 0; stats->matvecs = 0; stats->nconv = 0; stats->err_est = 0;
  if (tol <= 0) tol = 1e-8;
  if (x != y) DNM_TRY(vk_copy(y, x, n_local, st));
  if (t == 0.0) { stats->reason = DNM_CONVERGED_TOL; return 0; }
  double r = 0;
  DNM_TRY(dnm_mat_norm_inf(A, &r, stream));
  DNM_TRY(ops.maxr(&r, 1));
  if (hooks && hooks->allreduce_max) dnm_mat_set_norm(A, r);
  if (r == 0.0) { stats->reason = DNM_CONVERGED_TOL; return 0; }
  void *W = nullptr;
  DNM_TRY(basis_workspace((size_t)4 * (size_t)n_local * 16, &W));
  int nsteps = 0;
  DNM_TRY(cheb_core(ops, y, n_local, t, tol, r, W, &nsteps, &stats->err_est));
  DNM_HIP(hipStreamSynchronize(st));
  stats->reason = DNM_CONVERGED_TOL;
  stats->its = nsteps;
  stats->matvecs = ops.matvecs;
  return 0;
}

int dnm_expm_multiply(dnm_mat *A, const void *x, void *y, int64_t n_local, double scale_re,
                      double scale_im, double tol, int ncv, int max_its, size_t work_limit_bytes,
                      const dnm_hooks *hooks, dnm_solver_stats *stats, void *stream) {
  DNM_CHECK(A && x && y && stats, "null argument");
  DNM_CHECK(!A->real_packed, "exp(-iHt) needs complex vectors: not for a real-packed operator");
  hipStream_t st = (hipStream_t)stream;
  Ops ops{A, hooks, st, n_local};
  stats->reason = 0; stats->its = 0; stats->matvecs = 0; stats->nconv = 0; stats->err_est = 0;
  int64_t Nglob = A->N;
  // defaults everywhere and a real time: the driver may hand the rest of the interval to the Chebyshev expansion
  // (DNM_EXPM_HYBRID=0 keeps it Krylov throughout)
  const char *henv = knob("DNM_EXPM_HYBRID");
  const bool hybrid = ncv <= 0 && max_its <= 0 && scale_re == 0.0 && !(henv && henv[0] == '0');
  if (tol <= 0) tol = 1e-8;
  if (max_its <= 0) max_its = 100;
  int m = ncv > 0 ? ncv : 30;
  if ((int64_t)m > Nglob) m = (int)Nglob;
  if (work_limit_bytes) {
    // the limit is the caller's view of free device memory; the cached workspace is ours to reuse
    int64_t fit = (int64_t)((work_limit_bytes + g_basis.bytes) / ((size_t)n_local * 16)) - 2;
    if (fit < 2) fit = 2;
    if (m > fit) m = (int)fit;
  }
  {
    // a cached workspace of a useful size is reused as it is: growing it by a few vectors means handing back and
    // re-acquiring the whole slab, which the driver clears at ~30 GB/s (seconds at these sizes) -- more than a
    // slightly larger basis saves
    const int64_t have = (int64_t)(g_basis.bytes / ((size_t)n_local * 16));
    if (ncv <= 0 && have >= 12 && (int64_t)m + 2 > have) m = (int)(have - 2);
  }
  if (m < 1) m = 1;
  {
    // every rank sizes its basis from its own free memory and cached workspace: they must run the same m (the
    // all-reduce lengths and the multiply counts depend on it), so take the smallest
    double neg = -(double)m;
    DNM_TRY(ops.maxr(&neg, 1));
    m = (int)(-neg);
  }

  const zc scale(scale_re, scale_im);
  const double t_out = std::abs(scale);
  DNM_TRY(vk_copy(y, x, n_local, st));
  if (t_out == 0.0) { stats->reason = DNM_CONVERGED_TOL; return 0; }
  const zc dir = scale / t_out;

  double anorm = 0;
  DNM_TRY(dnm_mat_norm_inf(A, &anorm, stream));
  DNM_TRY(ops.maxr(&anorm, 1));
  if (hooks && hooks->allreduce_max) dnm_mat_set_norm(A, anorm);

  double beta = 0;
  DNM_TRY(ops.norm(y, &beta));
  if (beta == 0.0 || anorm == 0.0) { stats->reason = DNM_CONVERGED_TOL; return 0; }

  if (hybrid) {
    // skip the Krylov probe when the expansion -- whose term count is known exactly -- is the cheaper one: an
    // earlier solve with this operator ended in it and the Krylov step size seen then still says so for this
    // interval; or the whole expansion costs less than ONE outer Krylov step of m multiplies (short time steps:
    // no basis, and none of the seconds a 200 GiB workspace takes to acquire)
    int64_t terms = 0;
    DNM_TRY(cheb_cost(anorm * t_out, tol, &terms));
    bool go = false;
    if (A->expm_tstep > 0.0) {
      const double kry = 1.9 * (double)A->expm_m * std::ceil(t_out / A->expm_tstep);
      go = 1.25 * (double)terms < 0.8 * kry;
    }
    if (!go) go = 1.25 * (double)terms <= 1.9 * (double)m;
    void *W = nullptr;
    if (!go) {
      // the Krylov probe would have to acquire a large workspace first (seconds, see basis_workspace) for a basis
      // that memory keeps short -- where the expansion wins unless the norm bound is loose (m = 11 at L = 30: 144
      // Krylov multiplies against 56 terms for t = 1, i.e. the bound may exceed the spectral radius about
      // threefold before the expansion loses).  Ten Lanczos steps from x (the expansion's own four vectors
      // suffice) tell: their extreme Ritz values reach roughly half the radius (random-field Heisenberg chain:
      // 0.3 of the infinity norm; SYK at L = 8, where ten steps see all of it: 0.18), so 0.2 of the bound is the line.
      const double need = (double)(m + 2) * (double)n_local * 16.0;
      double want = (need > (double)g_basis.bytes && need >= 48.0 * 1073741824.0 && m < 30 && Nglob > 64) ? 1.0 : 0.0;
      DNM_TRY(ops.maxr(&want, 1));
      const char *penv = knob("DNM_EXPM_PROBE");
      if (penv) want = penv[0] == '1' && Nglob > 64 ? 1.0 : 0.0;
      if (want > 0.0 && A->expm_bound != 0) {       // probed before with this operator
        go = A->expm_bound > 0;
        want = 0.0;
      }
      if (want > 0.0) {
        DNM_TRY(basis_workspace((size_t)4 * (size_t)n_local * 16, &W));
        double rho = 0;
        DNM_TRY(lanczos_extent(ops, y, beta, n_local, W, 10, &rho, st));
        go = rho >= 0.2 * anorm;
        A->expm_bound = go ? 1 : -1;
        if (knob("DNM_KRYLOV_DEBUG"))
          fprintf(stderr, "dnm_expm_multiply: spectral extent seen by x %.4g of the bound %.4g -> %s\n", rho, anorm,
                  go ? "Chebyshev expansion" : "Krylov");
      }
    }
    if (go) {
      if (!W) DNM_TRY(basis_workspace((size_t)4 * (size_t)n_local * 16, &W));
      int csteps = 0;
      double cerr = 0;
      DNM_TRY(cheb_core(ops, y, n_local, -dir.imag() * t_out, tol, anorm, W, &csteps, &cerr));
      DNM_HIP(hipStreamSynchronize(st));
      stats->reason = DNM_CONVERGED_TOL;
      stats->its = csteps;
      stats->matvecs = ops.matvecs;
      stats->err_est = cerr;
      return 0;
    }
  }

  void *V = nullptr;   // v_0..v_m plus one scratch vector
  DNM_TRY(basis_workspace((size_t)(m + 2) * (size_t)n_local * 16, &V));
  void *tmpv = vecptr(V, n_local, m + 1);

  const double eps = 2.220446049250313e-16;
  const double rndoff = anorm * eps, break_tol = 1e-7, gamma = 0.9, delta = 1.2;
  const int mxrej = 10;
  double xm = 1.0 / m;
  double t_now = 0, s_error = 0;
  const double fact = std::pow((m + 1) / std::exp(1.0), m + 1) * std::sqrt(2.0 * M_PI * (m + 1));
  double t_new = (1.0 / anorm) * std::pow((fact * tol) / (4.0 * beta * anorm), xm);
  t_new = round2(t_new);

  const int mh = m + 2;
  std::vector<zc> H, F, Hs;
  int nstep = 0;
  // DNM_EXPM_ORTHO=full: orthogonalise every Krylov vector against the whole basis (what
  // SLEPc's BV does); default: Lanczos with partial re-orthogonalisation
  const char *oenv = knob("DNM_EXPM_ORTHO");
  const bool use_pro = !(oenv && oenv[0] == 'f');
  LanczosMonitor mon;
  while (t_now < t_out) {
    if (nstep >= max_its) {
      stats->reason = DNM_DIVERGED_ITS;
      stats->its = nstep; stats->matvecs = ops.matvecs; stats->err_est = s_error;
      return 0;
    }
    ++nstep;
    double t_step = std::min(t_out - t_now, t_new);
    H.assign((size_t)mh * mh, zc(0));
    int mb = m, k1 = 2;
    double avnorm = 0;
    std::vector<zc> h;
    // PRO path: the basis is stored UNNORMALISED, w_j = nv[j] v_j, so no vector is ever rescaled:
    //   w_{j+1} = H w_j - alpha_j w_j - (beta_j nv[j]/nv[j-1]) w_{j-1},  nv[j+1] = |w_{j+1}| = nv[j] beta_{j+1}
    // (one dot + one fused update per step); the scales enter every coefficient on the host.
    std::vector<double> nv(m + 2, 1.0), bet(m + 2, 0.0);
    if (use_pro) {
      mon.reset(m, (double)Nglob, tol);
      DNM_TRY(vk_copy(vecptr(V, n_local, 0), y, n_local, st));
      nv[0] = beta;
    } else {
      // v_0 = w / beta
      DNM_TRY(vk_axpby(vecptr(V, n_local, 0), y, n_local, 1.0 / beta, 0, 0, 0, st));
    }
    for (int j = 0; j < m; ++j) {
      void *p = vecptr(V, n_local, j + 1);
      zc d0(0);
      // the beta term of the recurrence rides on the multiply: p = H w_j - (beta_j nv_j / nv_{j-1}) w_{j-1}
      if (use_pro)
        DNM_TRY(ops.mult_dot(vecptr(V, n_local, j), p, &d0, j > 0 ? vecptr(V, n_local, j - 1) : nullptr,
                             j > 0 ? bet[j] * nv[j] / nv[j - 1] : 0.0));
      else DNM_TRY(ops.mult(vecptr(V, n_local, j), p));
      double hn = 0;
      if (use_pro) {
        const zc alpha = d0 / (nv[j] * nv[j]);
        h.assign(j + 1, zc(0));
        h[j] = alpha;
        if (j > 0) h[j - 1] = bet[j];
        double n2 = 0;
        DNM_TRY(vec_lanczos_update_host(p, vecptr(V, n_local, j), nullptr, n_local, alpha.real(), alpha.imag(), 0.0,
                                        &n2, st));
        DNM_TRY(ops.sum(&n2, 1));
        nv[j + 1] = std::sqrt(n2 > 0 ? n2 : 0.0);
        hn = nv[j + 1] / nv[j];
        if (mon.update(j, alpha.real(), hn)) {
          std::vector<zc> g, c(j + 1);
          DNM_TRY(ops.mdot(V, j + 1, p, g));            // g_i = <w_i, w_{j+1}>
          for (int i = 0; i <= j; ++i) {
            c[i] = -g[i] / (nv[i] * nv[i]);
            h[i] += g[i] / (nv[i] * nv[j]);
          }
          DNM_TRY(ops.maxpy(p, V, j + 1, c));
          double nn = 0;
          DNM_TRY(ops.norm(p, &nn));
          nv[j + 1] = nn;
          hn = nn / nv[j];
          mon.beta[j + 1] = hn;
        }
        bet[j + 1] = hn;
        if (hn > break_tol * anorm && (nv[j + 1] > 1e120 || nv[j + 1] < 1e-120)) {
          DNM_TRY(vk_scale(p, n_local, 1.0 / nv[j + 1], 0, st));     // keep the scales representable
          nv[j + 1] = 1.0;
        }
      } else {
        DNM_TRY(ops.orthogonalize(p, V, j + 1, h, &hn));
      }
      for (int i = 0; i <= j; ++i) H[(size_t)j * mh + i] = h[i];
      if (hn <= break_tol * anorm) {   // happy breakdown
        k1 = 0;
        mb = j + 1;
        t_step = t_out - t_now;
        break;
      }
      H[(size_t)j * mh + (j + 1)] = hn;
      if (!use_pro) DNM_TRY(vk_scale(p, n_local, 1.0 / hn, 0, st));
    }
    if (k1 != 0) {
      H[(size_t)m * mh + (m + 1)] = 1.0;
      // avnorm = || A v_m ||
      DNM_TRY(ops.mult(vecptr(V, n_local, m), tmpv));
      DNM_TRY(ops.norm(tmpv, &avnorm));
      if (use_pro) avnorm /= nv[m];
    }
    int ireject = 0;
    double err_loc = 0;
    int mx = mb + k1;
    while (true) {
      mx = mb + k1;
      Hs.assign((size_t)mx * mx, zc(0));
      for (int j = 0; j < mx; ++j)
        for (int i = 0; i < mx; ++i) Hs[(size_t)j * mx + i] = dir * t_step * H[(size_t)j * mh + i];
      DNM_CHECK(zexpm(mx, Hs, F) == 0, "dense expm failed");
      if (k1 == 0) { err_loc = break_tol; break; }
      const double p1 = std::abs(F[m]) * beta;
      const double p2 = std::abs(F[m + 1]) * beta * avnorm;
      if (p1 > 10.0 * p2) { err_loc = p2; xm = 1.0 / m; }
      else if (p1 > p2) { err_loc = (p1 * p2) / (p1 - p2); xm = 1.0 / m; }
      else { err_loc = p1; xm = 1.0 / std::max(1, m - 1); }
      if (err_loc <= delta * t_step * tol) break;
      t_step = gamma * t_step * std::pow(t_step * tol / err_loc, xm);
      t_step = round2(t_step);
      if (++ireject > mxrej) {
        stats->reason = DNM_DIVERGED_BREAKDOWN;
        stats->its = nstep; stats->matvecs = ops.matvecs; stats->err_est = s_error;
        return 0;
      }
    }
    // w = V[:, 0:mx') (beta F[0:mx', 0])
    const int mxw = mb + std::max(0, k1 - 1);
    std::vector<zc> c(mxw);
    for (int i = 0; i < mxw; ++i) c[i] = beta * F[i] / (use_pro ? nv[i] : 1.0);
    DNM_TRY(vk_set(y, n_local, 0, 0, st));
    DNM_TRY(ops.maxpy(y, V, mxw, c));
    DNM_TRY(ops.norm(y, &beta));
    t_now += t_step;
    t_new = gamma * t_step * std::pow(t_step * tol / err_loc, xm);
    t_new = round2(t_new);
    err_loc = std::max(err_loc, rndoff);
    s_error += err_loc;
    if (beta == 0.0) break;
    // With the caller's defaults (no ncv / max_its) and a real time, finish by the Chebyshev expansion when the
    // step size the error control has settled on makes that clearly cheaper: it costs ~1.25 multiply-times per
    // term against ~1.9 per Krylov multiply (measured), and its term count is known exactly.
    if (hybrid && m >= 2 && t_now < t_out) {
      const double t_left = t_out - t_now;
      int64_t terms = 0;
      DNM_TRY(cheb_cost(anorm * t_left, tol, &terms));
      const double kry = 1.9 * (double)m * std::ceil(t_left / t_new);
      if (1.25 * (double)terms < 0.8 * kry) {
        int csteps = 0;
        double cerr = 0;
        DNM_TRY(cheb_core(ops, y, n_local, -dir.imag() * t_left, tol, anorm, V, &csteps, &cerr));
        A->expm_tstep = t_new;
        A->expm_m = m;
        nstep += csteps;
        s_error += cerr;
        break;
      }
    }
  }
  DNM_HIP(hipStreamSynchronize(st));
  stats->reason = DNM_CONVERGED_TOL;
  stats->its = nstep;
  stats->matvecs = ops.matvecs;
  stats->err_est = s_error;
  return 0;
}

int dnm_eigsolve(dnm_mat *A, int64_t n_local, int nev, int which, double tol, int ncv, int max_its,
                 uint64_t seed, const dnm_hooks *hooks, int nev_max, double *evals, void *evecs,
                 dnm_solver_stats *stats, void *stream) {
  DNM_CHECK(A && evals && stats && nev >= 1 && nev_max >= nev, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  Ops ops{A, hooks, st, n_local};
  ops.real = A->real_packed;
  stats->reason = 0; stats->its = 0; stats->matvecs = 0; stats->nconv = 0; stats->err_est = 0;
  const int64_t Nglob = A->N;
  if (tol <= 0) tol = 1e-8;
  // ncv < 0: the default basis, but at most -ncv vectors in all (what fits in device memory: the caller's limit)
  int cap = 0;
  if (ncv < 0) { cap = -ncv; ncv = 0; }
  int m = ncv > 0 ? ncv : std::max(2 * nev, nev + 15);
  if (cap > 0 && m + 1 > cap) m = cap - 1;
  if ((int64_t)m > Nglob) m = (int)Nglob;
  {
    double neg = -(double)m;          // the same basis size on every rank (see dnm_expm_multiply)
    DNM_TRY(ops.maxr(&neg, 1));
    m = (int)(-neg);
  }
  // (several pairs whose restarted basis does not fit -- fewer than nev + 2 vectors beside the residual and the
  // filter's two work vectors -- go one after the other through the basis-free recurrence on the deflated
  // operator, see eigsolve_basis_free)
  const bool no_room = cap > 0 && cap < nev + 6;
  DNM_CHECK(m >= nev || no_room, "ncv smaller than nev");
  if (max_its <= 0) max_its = (int)std::min<int64_t>(std::max<int64_t>(100, 2 * Nglob / std::max(m, 1)), 1 << 30);
  bool filtered = false;
  {
    // one extremal pair of a large operator under default parameters: Lanczos without a stored basis (a step is
    // the multiply plus one sweep; the restarted scheme below spends two thirds of its time on basis traffic at
    // these sizes).  DNM_EIGS_BASISFREE=0 / 1 forces the choice; an explicit ncv keeps the restarted scheme.
    const char *bf = knob("DNM_EIGS_BASISFREE");
    double negn = -(double)n_local;       // the smallest block decides, so that every rank takes the same path
    DNM_TRY(ops.maxr(&negn, 1));
    const bool large = -negn >= (double)((int64_t)1 << 22);
    const bool want = bf ? bf[0] == '1' : large;
    const bool forced = bf && bf[0] == '1';
    if (ncv <= 0 && Nglob > 64 && Nglob > 4 * (int64_t)nev && ((want && nev == 1) || forced || no_room)) {
      const int64_t steps64 = std::min<int64_t>((int64_t)max_its * std::max(m, nev + 15), Nglob);
      return eigsolve_basis_free(ops, A, n_local, nev, which, tol, (int)std::min<int64_t>(steps64, 100000), seed,
                                 hooks, evals, evecs, stats, st);
    }
    DNM_CHECK(m >= nev, "not enough memory for a restarted basis");
    // several pairs at one end of the spectrum of a large operator: thick-restart Lanczos on a Chebyshev filter
    // p(H) -- d fused multiplies per Lanczos vector, so the orthogonalisation and restart traffic per multiply
    // drops by d (at 4-16 GiB per vector the plain scheme spends 80 % of its time there).  DNM_EIGS_FILTER=0 / 1
    // forces the choice; an explicit ncv keeps the plain scheme.
    const char *fe = knob("DNM_EIGS_FILTER");
    filtered = (fe ? fe[0] == '1' : (large && nev > 1)) && which != DNM_WHICH_EXTERIOR && ncv <= 0 &&
               Nglob > 8 * (int64_t)m;
    if (filtered && cap > 0 && m + 3 > cap) {        // the filter's two work vectors come out of the same budget
      if (cap - 3 >= nev + 2) m = cap - 3;
      else filtered = false;
    }
  }
  DNM_CHECK((size_t)(m + 1) * 16 <= 160 * 1024, "ncv too large for the basis-rotation kernel");

  void *V = nullptr;
  DNM_TRY(basis_workspace((size_t)(m + 1 + (filtered ? 2 : 0)) * (size_t)n_local * 16, &V));

  // start vector: counter-based normal deviates keyed by the global index (the rank's first row: blocks may be
  // uneven, PetscSplitOwnership), written in the vectors' layout
  const int64_t offset = hooks ? A->row0 : 0;
  DNM_TRY(random_start(A, vecptr(V, n_local, 0), n_local, seed, offset, st));
  double nrm0 = 0;
  DNM_TRY(ops.norm(vecptr(V, n_local, 0), &nrm0));
  DNM_CHECK(nrm0 > 0, "zero start vector");
  DNM_TRY(vk_scale(vecptr(V, n_local, 0), n_local, 1.0 / nrm0, 0, st));

  ChebFilter flt;
  int which_p = which;                     // the end of p(H)'s spectrum the wanted pairs sit at
  double tol_p = tol, tol_h = 0.5 * tol;   // on a filter: its own relative residual / the estimate of H's
  if (filtered) {
    // where to cut: a few steps of plain Lanczos (three rotating vectors) give Ritz values theta_i >= lambda_i
    // (from below at the other end) and the last residual norm; |H|_inf bounds the far end rigorously
    const int k0 = (int)std::min<int64_t>(Nglob - 1, std::max(40, 10 * nev));
    std::vector<double> al, be;
    auto slot = [&](int k) { return (void *)vecptr(V, n_local, k % 3); };
    for (int j = 0; j < k0; ++j) {
      void *q = slot(j), *pq = slot(j + 1), *qm = slot(j + 2);
      zc dd(0);
      double pn2 = 0;
      DNM_TRY(ops.mult_dot(q, pq, &dd, j > 0 ? qm : nullptr, j > 0 ? be[j - 1] : 0.0, &pn2));
      al.push_back(dd.real());
      double n2 = 0;
      DNM_TRY(vec_lanczos_update_host(pq, q, nullptr, n_local, dd.real(), dd.imag(), 0.0, &n2, st, 1.0));
      DNM_TRY(ops.sum(&n2, 1));
      const double bn = std::sqrt(n2 > 0 ? n2 : 0.0);
      be.push_back(bn);
      if (bn <= 1e-12 * (std::fabs(dd.real()) + 1.0)) break;        // invariant subspace: the plain scheme copes
      DNM_TRY(vk_scale(pq, n_local, 1.0 / bn, 0, st));
    }
    const int kk = (int)al.size();
    const int margin = std::max(2, (nev + 1) / 2);
    if (kk < nev + margin + 2) {
      filtered = false;
    } else {
      std::vector<double> Tm((size_t)kk * kk, 0.0), wv, Sv;
      for (int i = 0; i < kk; ++i) {
        Tm[(size_t)i * kk + i] = al[i];
        if (i + 1 < kk) Tm[(size_t)(i + 1) * kk + i] = Tm[(size_t)i * kk + i + 1] = be[i];
      }
      jacobi_eig(kk, Tm, wv, Sv);
      std::sort(wv.begin(), wv.end());
      double nrmH = 0;
      DNM_TRY(dnm_mat_norm_inf(A, &nrmH, (void *)st));
      DNM_TRY(ops.maxr(&nrmH, 1));
      const double blast = be[kk - 1];
      double a_cut, far, near_t, gam;
      if (which == DNM_WHICH_LOWEST) {
        far = std::min(nrmH, wv[kk - 1] + blast);
        a_cut = wv[nev + margin - 1];
        near_t = wv[nev - 1];
        gam = (a_cut - near_t) / (far - a_cut);
        flt.ref = wv[0];
        which_p = DNM_WHICH_LOWEST;           // odd degree: p < 0 below the interval, ordered as H
      } else {
        far = std::max(-nrmH, wv[0] - blast);
        a_cut = wv[kk - nev - margin];
        near_t = wv[kk - nev];
        gam = (near_t - a_cut) / (a_cut - far);
        flt.ref = wv[kk - 1];
        which_p = DNM_WHICH_HIGHEST;
      }
      if (!(gam > 1e-9) || !(std::fabs(far - a_cut) > 0)) {
        filtered = false;                      // no usable gap estimate (degenerate Ritz values): plain scheme
      } else {
        // Degree: the filter cannot separate the wanted values from EACH OTHER (their images stay as close,
        // relatively, as d times their distance in acosh), so that work stays with the outer Lanczos process at d
        // multiplies per vector: a strong filter needs fewer vectors but more multiplies in all.  With a step costing
        // d multiplies plus two orthogonalisation passes over the basis (about 7 multiply-times at m = 20) the
        // measured optimum is an amplification of the nev-th value of about cosh(3.3) = 14 over the damped interval
        // -- degree 9 for the chains at L = 26...30: L=28, nev=5, tol 1e-10: d = 5 / 9 / 13 / 17 / 21 take
        // 6.2 / 5.6 / 5.9 / 6.1 / 7.3 s (plain restarted scheme: 12.1 s; profiles/r03_exp5_eigs_degree.txt)
        int d = (int)std::ceil(3.3 / (2.0 * std::sqrt(gam)));
        d = std::max(5, std::min(d, 49));
        if (const char *de = knob("DNM_EIGS_FILTER_DEGREE")) d = std::max(1, atoi(de));
        d |= 1;
        flt.d = d;
        flt.c = 0.5 * (a_cut + far);
        flt.h = 0.5 * std::fabs(far - a_cut);
        flt.ta = vecptr(V, n_local, m + 1);
        flt.tb = vecptr(V, n_local, m + 2);
        flt.on = true;
        ops.flt = &flt;
        tol_p = 0.25 * tol;
        if (knob("DNM_KRYLOV_DEBUG"))
          fprintf(stderr, "dnm_eigsolve (filtered): %d probe steps, cut %.6g, far end %.6g (|H|_inf %.6g), nev-th estimate "
                  "%.6g, relative gap %.3g, degree %d\n", kk, a_cut, far, nrmH, near_t, gam, d);
      }
    }
    // the probe used the first three slots: the start vector again
    DNM_TRY(random_start(A, vecptr(V, n_local, 0), n_local, seed, offset, st));
    DNM_TRY(vk_scale(vecptr(V, n_local, 0), n_local, 1.0 / nrm0, 0, st));
  }
  std::vector<double> theta, spike;        // kept Ritz values and their coupling to v_l
  std::vector<double> alpha(m, 0.0), betav(m, 0.0);
  int l = 0, its = 0, nconv = 0;
  int extra_matvecs = 0;
  int nok = 0;                             // filtered: leading Ritz pairs whose MEASURED residual passes
  std::vector<double> rq;                  // ... and their Rayleigh quotients in H
  double worst_true = 0.0;
  std::vector<double> T, w, Sm;
  std::vector<int> order(m);
  std::vector<zc> h;
  double anorm_est = 0;

  // DNM_EIGS_ORTHO=full: orthogonalise every Lanczos vector against the whole basis (what SLEPc's
  // Krylov-Schur does); default: partial re-orthogonalisation driven by the omega-recurrence
  const char *oenv = knob("DNM_EIGS_ORTHO");
  // (on a filter the basis work is a small share of a step and p(H) has a huge dynamic range: every step in full)
  const bool use_pro = !(oenv && oenv[0] == 'f') && (!filtered || knob("DNM_EIGS_FILTER_PRO") != nullptr);
  // DNM_EIGS_BETA=sweep: beta from a norm sweep after the update (never the fused form); =rescale: always run the
  // corrective rescaling sweep -- both only to exercise the rarely taken branches in tests
  const bool known_off = knob("DNM_EIGS_KNOWN") && knob("DNM_EIGS_KNOWN")[0] == '0';   // A/B switch
  const char *benv = knob("DNM_EIGS_BETA");
  const int beta_mode = !benv ? 0 : (benv[0] == 's' ? 1 : (benv[0] == 'r' ? 2 : 0));
  RestartMonitor mon;
  mon.init(m, (double)Nglob, filtered ? tol_p : tol);
  std::vector<double> row_l;
  while (true) {
    ++its;
    if (use_pro) mon.begin_cycle(l, theta, spike, row_l);
    for (int j = l; j < m; ++j) {
      void *p = vecptr(V, n_local, j + 1);
      const bool three_term = use_pro && j != l;
      zc d0(0);
      double pn2 = 0;
      if (three_term)
        DNM_TRY(ops.mult_dot(vecptr(V, n_local, j), p, &d0, vecptr(V, n_local, j - 1), betav[j - 1], &pn2));
      else DNM_TRY(ops.mult(vecptr(V, n_local, j), p));
      double bn = 0;
      bool normalised = false;
      if (!three_term) {
        // the first step of a cycle removes the spike components: whole basis, twice when Ritz vectors are
        // present (they are orthonormal to sqrt(eps) only under partial re-orthogonalisation)
        if (use_pro && l > 0 && j == l && (int)spike.size() == l && !known_off)
          DNM_TRY(ops.orthogonalize_known(p, V, j + 1, spike, h, &bn));
        else
          DNM_TRY(ops.orthogonalize(p, V, j + 1, h, &bn, (use_pro && l > 0) ? 2 : 1));
        alpha[j] = h[j].real();
        if (use_pro) mon.first_step(alpha[j], bn);
      } else {
        alpha[j] = d0.real();
        // |p - alpha q_j|^2 = |p|^2 - |alpha|^2 (q_j has unit norm): beta is known before the update sweep, which
        // then writes q_{j+1} = (p - alpha q_j) / beta directly; its own sum of squares (1 up to the cancellation
        // in the difference) corrects beta and, if it is off, the vector
        const double b2 = pn2 - std::norm(d0);
        const bool fused = b2 > 1e-4 * pn2 && pn2 > 0 && beta_mode != 1;
        const double best = fused ? std::sqrt(b2) : 0.0;
        const bool reorth = fused ? mon.update(j, alpha[j], best) : false;
        double n2 = 0;
        DNM_TRY(vec_lanczos_update_host(p, vecptr(V, n_local, j), nullptr, n_local, d0.real(), d0.imag(), 0.0, &n2,
                                        st, (fused && !reorth) ? 1.0 / best : 1.0));
        DNM_TRY(ops.sum(&n2, 1));
        if (fused && !reorth) {
          const double nu = std::sqrt(n2 > 0 ? n2 : 0.0);
          bn = best * nu;
          mon.beta[j + 1] = bn;
          normalised = true;
          if ((std::fabs(n2 - 1.0) > 1e-12 || beta_mode == 2) && nu > 0)
            DNM_TRY(vk_scale(p, n_local, 1.0 / nu, 0, st));
        } else {
          bn = std::sqrt(n2 > 0 ? n2 : 0.0);
          if (fused) mon.beta[j + 1] = bn;
          if (reorth || (!fused && mon.update(j, alpha[j], bn))) {
            std::vector<zc> g, c(j + 1);
            DNM_TRY(ops.mdot(V, j + 1, p, g));
            for (int i = 0; i <= j; ++i) c[i] = -g[i];
            DNM_TRY(ops.maxpy(p, V, j + 1, c));
            DNM_TRY(ops.norm(p, &bn));
            mon.beta[j + 1] = bn;
          }
        }
      }
      betav[j] = bn;
      anorm_est = std::max(anorm_est, std::fabs(alpha[j]) + bn);
      if (bn <= 1e-14 * std::max(1.0, anorm_est)) {
        // invariant subspace: continue with a fresh direction orthogonal to the basis
        betav[j] = 0.0;
        DNM_TRY(random_start(A, p, n_local, seed + 7919u * (uint64_t)(its * m + j + 1), offset, st));
        double rn = 0;
        DNM_TRY(ops.orthogonalize(p, V, j + 1, h, &rn, 2));
        DNM_CHECK(rn > 0, "Lanczos breakdown: could not extend the basis");
        DNM_TRY(vk_scale(p, n_local, 1.0 / rn, 0, st));
        if (use_pro) {
          mon.beta[j + 1] = 0.0;
          for (int k = 0; k <= j; ++k) mon.wcur[k] = mon.eps1;
        }
      } else if (!normalised) {
        DNM_TRY(vk_scale(p, n_local, 1.0 / bn, 0, st));
      }
    }
    // projected matrix: diag(theta) + spike row/col at l, tridiagonal beyond
    T.assign((size_t)m * m, 0.0);
    for (int i = 0; i < l; ++i) {
      T[(size_t)i * m + i] = theta[i];
      T[(size_t)l * m + i] = T[(size_t)i * m + l] = spike[i];
    }
    for (int j = l; j < m; ++j) {
      T[(size_t)j * m + j] = alpha[j];
      if (j + 1 < m) T[(size_t)(j + 1) * m + j] = T[(size_t)j * m + (j + 1)] = betav[j];
    }
    jacobi_eig(m, T, w, Sm);
    for (int i = 0; i < m; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
      if (which_p == DNM_WHICH_LOWEST) return w[a] < w[b];
      if (which_p == DNM_WHICH_HIGHEST) return w[a] > w[b];
      return std::fabs(w[a]) > std::fabs(w[b]);
    });
    const double bm = betav[m - 1];
    nconv = 0;
    for (int i = 0; i < m; ++i) {
      const int c = order[i];
      const double res = std::fabs(bm * Sm[(size_t)c * m + (m - 1)]);
      // relative to the eigenvalue (SLEPc EPS_CONV_REL); on a filter: the residual as H would see it (the measured
      // residual of the restarted vectors has the last word)
      bool ok = res <= (filtered ? tol_p : tol) * std::max(std::fabs(w[c]), 1e-300);
      if (filtered && !ok) {
        double lam;
        ok = flt.seen_from_a(w[c], res, which == DNM_WHICH_LOWEST, &lam) <= tol_h;
      }
      if (ok) ++nconv; else break;
    }
    const bool stop = nconv >= nev || its >= max_its;
    if (stop && !filtered) break;
    // thick restart: keep the converged pairs plus half of the rest
    int keep = nconv + std::max(1, (m - nconv) / 2);
    if (filtered && keep < nev) keep = nev;
    if (keep > m - 1) keep = m - 1;
    std::vector<double> Ssel((size_t)2 * m * keep, 0.0);
    theta.assign(keep, 0.0);
    spike.assign(keep, 0.0);
    for (int o = 0; o < keep; ++o) {
      const int c = order[o];
      theta[o] = w[c];
      spike[o] = bm * Sm[(size_t)c * m + (m - 1)];
      for (int j = 0; j < m; ++j) Ssel[2 * ((size_t)o * m + j)] = Sm[(size_t)c * m + j];
    }
    const double *sd = nullptr;
    DNM_TRY(vec_upload_coefs(Ssel.data(), Ssel.size(), st, &sd));
    DNM_TRY(vk_basis_update(V, n_local, m, keep, n_local, sd, st));
    DNM_TRY(vk_copy(vecptr(V, n_local, keep), vecptr(V, n_local, m), n_local, st));
    if (use_pro) {
      // |q_m^H u_o| <= sum_k |S_ko| |omega_{m,k}|: the new q_l against the rotated basis
      row_l.assign(keep, 0.0);
      for (int o = 0; o < keep; ++o)
        for (int k = 0; k < m; ++k) row_l[o] += std::fabs(Sm[(size_t)order[o] * m + k]) * std::fabs(mon.wcur[k]);
    }
    l = keep;
    if (filtered && stop) {
      // the filter's estimates say the wanted pairs have converged (or the iteration limit is reached): the kept
      // Ritz vectors now sit in the first slots -- measure what the contract is about, |H u - <u,Hu> u| / |<u,Hu>|
      // in H itself (one multiply and one sweep each; the filter's work vectors are free in between)
      const int nchk = std::min(std::min(keep, nev_max), std::max(nev, nconv));
      flt.on = false;
      const int mv0 = ops.matvecs;
      rq.assign(nchk, 0.0);
      nok = 0;
      worst_true = 0.0;
      bool chain = true;
      for (int o = 0; o < nchk; ++o) {
        void *u = vecptr(V, n_local, o), *hu = flt.ta;
        zc dd(0);
        DNM_TRY(ops.mult_dot(u, hu, &dd));
        double n2 = 0;
        DNM_TRY(vec_lanczos_update_host(hu, u, nullptr, n_local, dd.real(), dd.imag(), 0.0, &n2, st));
        DNM_TRY(ops.sum(&n2, 1));
        rq[o] = dd.real();
        const double rel = std::sqrt(n2 > 0 ? n2 : 0.0) / std::max(std::fabs(rq[o]), 1e-300);
        if (chain && rel <= tol) { ++nok; worst_true = std::max(worst_true, rel); }
        else {
          if (chain && o < nev) worst_true = std::max(worst_true, rel);
          chain = false;
        }
      }
      extra_matvecs += ops.matvecs - mv0;
      ops.matvecs = mv0;
      flt.on = true;
      if (knob("DNM_KRYLOV_DEBUG"))
        fprintf(stderr, "dnm_eigsolve (filtered): restart %d, %d pairs converged on the filter (tol %.1e, estimate for H "
                "%.1e), %d pass in H (worst of the wanted %.2e)\n", its, nconv, tol_p, tol_h, nok, worst_true);
      if (nok >= nev || its >= max_its) break;
      // the estimates were satisfied too early: ask for more, by what the measurement missed
      const double f = std::max(1e-3, std::min(0.3, 0.3 * tol / std::max(worst_true, 1e-300)));
      tol_p *= f;
      tol_h *= f;
    }
  }

  if (filtered) {
    // the Ritz vectors are in place (thick restart) and measured in H: ordered by their Rayleigh quotients
    flt.on = false;
    const int nout = std::min(nok, nev_max);
    std::vector<int> ord(nout);
    for (int i = 0; i < nout; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return which == DNM_WHICH_LOWEST ? rq[a] < rq[b] : rq[a] > rq[b]; });
    for (int i = 0; i < nout; ++i) evals[i] = rq[ord[i]];
    if (evecs)
      for (int i = 0; i < nout; ++i)
        DNM_TRY(vk_copy((char *)evecs + (size_t)i * (size_t)n_local * 16, vecptr(V, n_local, ord[i]), n_local, st));
    if (knob("DNM_KRYLOV_DEBUG"))
      fprintf(stderr, "dnm_eigsolve (filtered): %d restarts, %d matvecs (+%d for the checks), degree %d, largest true "
              "relative residual %.2e\n", its, ops.matvecs, extra_matvecs, flt.d, worst_true);
    DNM_HIP(hipStreamSynchronize(st));
    stats->its = its;
    stats->matvecs = ops.matvecs;
    stats->nconv = nout;
    stats->err_est = worst_true;
    stats->reason = (nout >= nev) ? DNM_CONVERGED_TOL : DNM_DIVERGED_ITS;
    return 0;
  }
  const int nout = std::min(nconv, nev_max);
  for (int i = 0; i < nout; ++i) evals[i] = w[order[i]];
  if (nout > 0) {
    std::vector<double> Ssel((size_t)2 * m * nout, 0.0);
    for (int o = 0; o < nout; ++o)
      for (int j = 0; j < m; ++j) Ssel[2 * ((size_t)o * m + j)] = Sm[(size_t)order[o] * m + j];
    const double *sd = nullptr;
    DNM_TRY(vec_upload_coefs(Ssel.data(), Ssel.size(), st, &sd));
    DNM_TRY(vk_basis_update(V, n_local, m, nout, n_local, sd, st));
    if (use_pro)     // a semi-orthogonal basis leaves the Ritz vectors orthonormal to sqrt(eps) only: Gram-Schmidt
      for (int o = 0; o < nout; ++o) {   // (inside a degenerate level the gap argument does not protect them)
        double nn = 0;
        if (o > 0) DNM_TRY(ops.orthogonalize(vecptr(V, n_local, o), V, o, h, &nn));
        else DNM_TRY(ops.norm(vecptr(V, n_local, o), &nn));
        DNM_CHECK(nn > 0, "zero Ritz vector");
        DNM_TRY(vk_scale(vecptr(V, n_local, o), n_local, 1.0 / nn, 0, st));
      }
    if (evecs) DNM_TRY(vk_copy(evecs, V, (int64_t)nout * n_local, st));
    // what was promised, measured: the largest relative residual |H u - <u,Hu> u| / |theta| of the returned
    // pairs (one multiply each; the Lanczos vector in the last slot is no longer needed)
    double worst = 0.0;
    const int matvecs_solve = ops.matvecs;
    for (int o = 0; o < nout; ++o) {
      void *u = vecptr(V, n_local, o), *hu = vecptr(V, n_local, m);
      if (nout > m) break;
      zc d(0);
      DNM_TRY(ops.mult_dot(u, hu, &d));
      double n2 = 0;
      DNM_TRY(vec_lanczos_update_host(hu, u, nullptr, n_local, d.real(), d.imag(), 0.0, &n2, st));
      DNM_TRY(ops.sum(&n2, 1));
      worst = std::max(worst, std::sqrt(n2 > 0 ? n2 : 0.0) / std::max(std::fabs(evals[o]), 1e-300));
    }
    ops.matvecs = matvecs_solve;      // reported separately from the iteration's multiplies
    stats->err_est = worst;
  }
  if (knob("DNM_KRYLOV_DEBUG"))
    fprintf(stderr, "dnm_eigsolve: %d restarts, %d matvecs, %d three-term steps, %d full re-orthogonalisations, "
            "largest true relative residual %.2e\n", its, ops.matvecs, mon.steps, mon.reorths, stats->err_est);
  DNM_HIP(hipStreamSynchronize(st));
  stats->its = its;
  stats->matvecs = ops.matvecs;
  stats->nconv = nout;
  stats->reason = (nconv >= nev) ? DNM_CONVERGED_TOL : DNM_DIVERGED_ITS;
  return 0;
}

}  // extern "C"
