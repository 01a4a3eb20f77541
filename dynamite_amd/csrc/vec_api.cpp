// C ABI for the vector kernels (PETSc Vec / SLEPc BV replacements).
#include "vec_api.h"
#include "sc3.h"

#include <vector>

namespace dnm {

static DevBuf g_scratch;       // reduction partials + results
static DevBuf g_coef;          // small host->device coefficient uploads

int vec_scratch(size_t bytes, double **p) {
  if (g_scratch.bytes < bytes) DNM_TRY(g_scratch.alloc(bytes));
  *p = (double *)g_scratch.p;
  return 0;
}

int vec_upload_coefs(const double *host, size_t ndoubles, hipStream_t st, const double **dev) {
  if (g_coef.bytes < ndoubles * 8) DNM_TRY(g_coef.alloc(ndoubles * 8 > 65536 ? ndoubles * 8 : 65536));
  // the previous use of the buffer must be complete before we overwrite it
  DNM_HIP(hipStreamSynchronize(st));
  DNM_HIP(hipMemcpyAsync(g_coef.p, host, ndoubles * 8, hipMemcpyHostToDevice, st));
  DNM_HIP(hipStreamSynchronize(st));   // host buffer may be freed by the caller
  *dev = (const double *)g_coef.p;
  return 0;
}

int vec_mdot_host(const void *V, int64_t ldv, int nv, const void *w, int64_t n, double *h_host,
                  hipStream_t st) {
  DNM_CHECK(nv >= 1 && nv <= 256, "mdot: nv out of range");
  const int nb = vk_mdot_blocks(n);
  double *part = nullptr;
  DNM_TRY(vec_scratch(((size_t)nb + 1) * 2 * nv * sizeof(double), &part));
  DNM_TRY(vk_mdot(V, ldv, nv, w, n, part, st));
  DNM_HIP(hipMemcpyAsync(h_host, part + (size_t)nb * 2 * nv, (size_t)2 * nv * sizeof(double),
                         hipMemcpyDeviceToHost, st));
  DNM_HIP(hipStreamSynchronize(st));
  return 0;
}

int vec_lanczos_dot_host(void *y, const void *z, const void *x, int64_t n, double b, double *out3_host,
                         hipStream_t st, double yscale) {
  const int nb = vk_sweep_blocks(n);
  double *part = nullptr;
  DNM_TRY(vec_scratch(vk_sweep_scratch(n, 3) * sizeof(double), &part));
  DNM_TRY(vk_lanczos_dot(y, z, x, n, b, part, st, yscale));
  DNM_HIP(hipMemcpyAsync(out3_host, part + 3 * (size_t)nb, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
  DNM_HIP(hipStreamSynchronize(st));
  return 0;
}

int vec_lanczos_update_host(void *p, const void *v, const void *u, int64_t n, double are, double aim, double b,
                            double *norm2_host, hipStream_t st, double scale) {
  const int nb = vk_sweep_blocks(n);
  double *part = nullptr;
  DNM_TRY(vec_scratch(vk_sweep_scratch(n, 1) * sizeof(double), &part));
  DNM_TRY(vk_lanczos_update(p, v, u, n, are, aim, b, scale, part, st));
  DNM_HIP(hipMemcpyAsync(norm2_host, part + nb, sizeof(double), hipMemcpyDeviceToHost, st));
  DNM_HIP(hipStreamSynchronize(st));
  return 0;
}

}  // namespace dnm

using namespace dnm;
static hipStream_t S(void *s) { return (hipStream_t)s; }

extern "C" {

int dnm_vec_set(void *x, int64_t n, double re, double im, void *stream) {
  DNM_CHECK(x && n >= 0, "bad vector");
  return vk_set(x, n, re, im, S(stream));
}

int dnm_vec_copy(const void *x, void *y, int64_t n, void *stream) {
  DNM_CHECK(x && y && n >= 0, "bad vector");
  DNM_TRY(vk_copy(y, x, n, S(stream)));
  return 0;
}

int dnm_vec_scale(void *x, int64_t n, double re, double im, void *stream) {
  DNM_CHECK(x && n >= 0, "bad vector");
  return vk_scale(x, n, re, im, S(stream));
}

int dnm_vec_axpby(void *y, const void *x, int64_t n, double are, double aim, double bre, double bim,
                  void *stream) {
  DNM_CHECK(x && y && n >= 0, "bad vector");
  DNM_CHECK(x != y, "x and y cannot be the same vector");
  return vk_axpby(y, x, n, are, aim, bre, bim, S(stream));
}

int dnm_vec_dot(const void *x, const void *y, int64_t n, double *out, void *stream) {
  DNM_CHECK(x && y && out, "bad argument");
  // VecDot(x, y) = sum_i x_i conj(y_i) = (y^H x)
  return vec_mdot_host(y, n, 1, x, n, out, S(stream));
}

int dnm_vec_norm2(const void *x, int64_t n, double *out, void *stream) {
  DNM_CHECK(x && out, "bad argument");
  double h[2];
  DNM_TRY(vec_mdot_host(x, n, 1, x, n, h, S(stream)));
  *out = sqrt(h[0] > 0 ? h[0] : 0.0);
  return 0;
}

int dnm_vec_set_random_swz(void *x, int64_t n, uint64_t seed, int64_t offset, int swizzle, void *stream) {
  DNM_CHECK(x || n == 0, "null vector");
  DNM_CHECK(swizzle == 0 || (swizzle >= 5 && swizzle <= 24), "swizzle shift %d out of range", swizzle);
  return vk_random(x, n, seed, offset, S(stream), swizzle);
}

int dnm_vec_unpack_real(void *dst, const void *src, int64_t n_packed, int swizzle_packed, int swizzle_out, void *stream) {
  DNM_CHECK(dst && src && dst != src && n_packed >= 0, "bad vector");
  for (int sw : {swizzle_packed, swizzle_out})
    DNM_CHECK(sw == 0 || (sw >= 5 && sw <= 24), "swizzle shift %d out of range", sw);
  return vk_unpack_real(dst, src, n_packed, swizzle_packed, swizzle_out, S(stream));
}

int dnm_vec_swizzle_copy(void *dst, const void *src, int64_t n, int swizzle, void *stream) {
  DNM_CHECK((dst && src) || n == 0, "null vector");
  DNM_CHECK(dst != src, "dnm_vec_swizzle_copy works out of place");
  DNM_CHECK(swizzle == 0 || (swizzle >= 5 && swizzle <= 24), "swizzle shift %d out of range", swizzle);
  return vk_swizzle_copy(dst, src, n, swizzle, S(stream));
}

// ---- SpinConserve internal layout (sc3.h) ----------------------------------------------------------------
static const Sc3Layout *layout_of(const dnm_subspace *s, bool device) {
  if (!s || s->type != DNM_SPIN_CONSERVE || s->vec_swizzle == 0) {
    set_error("not a SpinConserve subspace with an internal vector layout");
    return nullptr;
  }
  const int a = sc3_code_a(s->vec_swizzle), w = sc3_code_w(s->vec_swizzle);
  if (!sc3_valid((int)s->L, (int)s->k, a, w)) {
    set_error("no such vector layout: L=%d k=%d a=%d w=%d", (int)s->L, (int)s->k, a, w);
    return nullptr;
  }
  return sc3_get((int)s->L, (int)s->k, a, w, device, sc3_code_order(s->vec_swizzle));
}

// the descriptor's site relabelling
static int perm_of(const dnm_subspace *s, Sc3Perm *P) {
  DNM_CHECK(sc3_perm_make(s->site_perm, (int)s->L, P), "site_perm is not a permutation of the %d spins", (int)s->L);
  return 0;
}

// the T blocks of a partition's rank (null: one rank)
static int part_range(const Sc3Layout &ly, const dnm_partition *part, uint32_t *T0, uint32_t *T1) {
  *T0 = 0;
  *T1 = 0xffffffffu;
  if (!part || part->nranks <= 1) return 0;
  DNM_CHECK(part->rank >= 0 && part->rank < part->nranks, "bad partition (rank %d of %d)", part->rank, part->nranks);
  const std::vector<uint32_t> Tb = sc3_partition(ly, part->nranks);
  *T0 = Tb[part->rank];
  *T1 = Tb[part->rank + 1];
  return 0;
}

int dnm_vec_layout_size(const dnm_subspace *s, int64_t *n) {
  DNM_CHECK(n, "null argument");
  const Sc3Layout *ly = layout_of(s, false);
  if (!ly) return 1;
  *n = ly->host.nint;
  return 0;
}

int dnm_vec_layout_partition(const dnm_subspace *s, int nranks, int rank, int64_t *istart, int64_t *ilen,
                             int64_t *nstart, int64_t *nlen) {
  DNM_CHECK(istart && ilen && nstart && nlen && nranks >= 1 && rank >= 0 && rank < nranks, "bad argument");
  const Sc3Layout *ly = layout_of(s, false);
  if (!ly) return 1;
  const std::vector<uint32_t> Tb = sc3_partition(*ly, nranks);
  sc3_range(*ly, Tb[rank], Tb[rank + 1], istart, ilen, nstart, nlen);
  return 0;
}

int dnm_vec_layout_blocks(const dnm_subspace *s, int64_t max, int64_t *T, int64_t *ibase, int64_t *count) {
  DNM_CHECK(count && max >= 0, "bad argument");
  const Sc3Layout *ly = layout_of(s, false);
  if (!ly) return 1;
  const int64_t n = (int64_t)ly->tseq.size();
  *count = n;
  if (max < n) return 0;
  DNM_CHECK(T && ibase, "null arrays");
  for (int64_t j = 0; j < n; ++j) {
    T[j] = ly->tseq[j];
    ibase[j] = ly->ibase[ly->tseq[j]];
  }
  ibase[n] = ly->host.nint;
  return 0;
}

int dnm_vec_layout_copy(const dnm_subspace *s, const dnm_partition *part, void *dst, const void *src, int to_internal,
                        void *stream) {
  DNM_CHECK(dst && src && dst != src, "dnm_vec_layout_copy works out of place on non-null vectors");
  const Sc3Layout *ly = layout_of(s, true);
  if (!ly) return 1;
  uint32_t T0, T1;
  DNM_TRY(part_range(*ly, part, &T0, &T1));
  Sc3Perm P;
  DNM_TRY(perm_of(s, &P));
  return sc3_layout_copy(*ly, dst, src, to_internal != 0, S(stream), T0, T1, &P);
}

int dnm_vec_layout_copy_f64(const dnm_subspace *s, const dnm_partition *part, double *dst, const double *src,
                            int to_internal, void *stream) {
  DNM_CHECK(dst && src && dst != src, "dnm_vec_layout_copy_f64 works out of place on non-null arrays");
  const Sc3Layout *ly = layout_of(s, true);
  if (!ly) return 1;
  uint32_t T0, T1;
  DNM_TRY(part_range(*ly, part, &T0, &T1));
  Sc3Perm P;
  DNM_TRY(perm_of(s, &P));
  return sc3_layout_copy_f64(*ly, dst, src, to_internal != 0, S(stream), T0, T1, &P);
}

int dnm_vec_layout_zero_padding(const dnm_subspace *s, const dnm_partition *part, void *x, void *stream) {
  const Sc3Layout *ly = layout_of(s, true);
  if (!ly) return 1;
  uint32_t T0, T1;
  DNM_TRY(part_range(*ly, part, &T0, &T1));
  if (T0 >= T1) return 0;          // a rank that owns no block (an empty tensor has a null data pointer)
  DNM_CHECK(x, "null vector");
  return sc3_zero_padding(*ly, x, S(stream), T0, T1);
}

int dnm_vec_layout_positions(const dnm_subspace *s, const dnm_partition *part, int64_t n, const int64_t *idx,
                             int64_t *pos, void *stream) {
  DNM_CHECK(n == 0 || (idx && pos), "null argument");
  const Sc3Layout *ly = layout_of(s, true);
  if (!ly) return 1;
  uint32_t T0, T1;
  DNM_TRY(part_range(*ly, part, &T0, &T1));
  Sc3Perm P;
  DNM_TRY(perm_of(s, &P));
  return sc3_positions(*ly, n, idx, pos, S(stream), T0, T1, &P);
}

int dnm_vec_layout_positions_host(const dnm_subspace *s, const dnm_partition *part, int64_t n, const int64_t *idx,
                                  int64_t *pos) {
  DNM_CHECK(n == 0 || (idx && pos), "null argument");
  const Sc3Layout *ly = layout_of(s, false);
  if (!ly) return 1;
  uint32_t T0, T1;
  DNM_TRY(part_range(*ly, part, &T0, &T1));
  int64_t is, il, ns, nl;
  sc3_range(*ly, T0, T1, &is, &il, &ns, &nl);
  SubView v{};
  v.type = DNM_SPIN_CONSERVE;
  v.L = ly->host.L;
  v.k = ly->host.k;
  v.ld = ly->host.L + 1;
  v.nchoosek = ly->host.nck;
  Sc3Perm P;
  DNM_TRY(perm_of(s, &P));
  // (whole vectors, or the half whose top bit is clear -- an XParity vector on top of the layout: both start at position 0
  // of the layout and at index 0 of the reference order, as perm_whole of the device paths accepts them)
  DNM_CHECK(!P.on || (is == 0 && ns == 0), "a relabelled SpinConserve layout is not partitioned over ranks");
  for (int64_t i = 0; i < n; ++i) {
    DNM_CHECK(idx[i] >= 0 && idx[i] < nl, "index %lld out of range", (long long)idx[i]);
    uint64_t st = (uint64_t)Sub<DNM_SPIN_CONSERVE>::i2s(idx[i] + ns, v);
    if (P.on) st = sc3_permute(st, P.to_int, P.L);
    pos[i] = sc3_pos(st, ly->host) - is;
  }
  return 0;
}

int dnm_vec_layout_unpack_real(const dnm_subspace *s, const dnm_partition *part, void *dst, const void *src, void *stream) {
  const Sc3Layout *ly = layout_of(s, true);
  if (!ly) return 1;
  uint32_t T0, T1;
  DNM_TRY(part_range(*ly, part, &T0, &T1));
  if (T0 >= T1) return 0;
  DNM_CHECK(dst && src && dst != src, "dnm_vec_layout_unpack_real works out of place on non-null arrays");
  return sc3_unpack_real(*ly, dst, (const double *)src, S(stream), T0, T1);
}

int dnm_vec_layout_set_random(const dnm_subspace *s, const dnm_partition *part, void *x, uint64_t seed, void *stream) {
  const Sc3Layout *ly = layout_of(s, true);
  if (!ly) return 1;
  uint32_t T0, T1;
  DNM_TRY(part_range(*ly, part, &T0, &T1));
  if (T0 >= T1) return 0;          // a rank that owns no block
  DNM_CHECK(x, "null vector");
  Sc3Perm P;
  DNM_TRY(perm_of(s, &P));
  return sc3_random(*ly, x, seed, S(stream), T0, T1, &P);
}

int dnm_sc_choose_site_perm(int L, int a, int w, int64_t nmasks, const int64_t *masks, int fix_top, int8_t *site_perm,
                            int32_t *counts) {
  DNM_CHECK(site_perm && (nmasks == 0 || masks), "null argument");
  DNM_CHECK(L >= 1 && L <= 63 && a >= 1 && w >= 1 && L - a - w >= 1, "no such layout: L=%d a=%d w=%d", L, a, w);
  sc3_choose_perm(L, a, w, nmasks, masks, fix_top != 0, site_perm, counts);
  return 0;
}

int dnm_vec_set_random(void *x, int64_t n, uint64_t seed, int64_t offset, void *stream) {
  DNM_CHECK(x && n >= 0, "bad vector");
  return vk_random(x, n, seed, offset, S(stream));
}

int dnm_vec_mdot(const void *V, int64_t ldv, int nv, const void *w, int64_t n, double *h_host,
                 void *stream) {
  DNM_CHECK(V && w && h_host, "bad argument");
  return vec_mdot_host(V, ldv, nv, w, n, h_host, S(stream));
}

int dnm_vec_maxpy(void *w, const void *V, int64_t ldv, int nv, int64_t n, const double *c_host,
                  void *stream) {
  DNM_CHECK(V && w && c_host && nv >= 1, "bad argument");
  const double *cd = nullptr;
  DNM_TRY(vec_upload_coefs(c_host, (size_t)2 * nv, S(stream), &cd));
  return vk_maxpy(w, V, ldv, nv, n, cd, S(stream));
}

int dnm_vec_basis_update(void *V, int64_t ldv, int nin, int nout, int64_t n, const double *S_host,
                         void *stream) {
  DNM_CHECK(V && S_host, "bad argument");
  const double *sd = nullptr;
  DNM_TRY(vec_upload_coefs(S_host, (size_t)2 * nin * nout, S(stream), &sd));
  return vk_basis_update(V, ldv, nin, nout, n, sd, S(stream));
}

}  // extern "C"
