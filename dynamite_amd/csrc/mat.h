// The shell-matrix handle behind the C ABI (replaces PETSc MatShell +
// shell_context, src/dynamite/_backend/shell_context.h:12-27).
#pragma once

#include <memory>
#include <vector>

#include "kernels.h"
#include "plan.h"
#include "sc3.h"
#include "subspace.h"

namespace dnm {

// Device allocation that frees itself.
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  int alloc(size_t nbytes);
  int upload(const void *host, size_t nbytes);
  void release();
};

// A subspace with owned host tables and their device mirrors.
struct SubOwned {
  SubView host{};
  SubView dev{};
  std::vector<int64_t> nck, smap, rind, rstates, bucket;
  DevBuf d_nck, d_smap, d_rind, d_rstates, d_bucket;
  int init(const dnm_subspace *s, bool want_device);
};

struct PassOnDevice {
  DevPass desc{};
  std::vector<DevQuad> h_quads;   // host copy (diagnostics / host-only handles)
  DevBuf quads;
  std::vector<double> h_dtile;    // in-tile diagonal per tile coordinate (DevPass::dtile), host copy
  DevBuf dtile;
  std::vector<DevTab> h_tabs;     // table records (plan.h: DevTab) and their tables, host copies
  std::vector<double> h_tabvals;
  DevBuf tabs, tabvals;
  int partner = -1;
  int n_eff = 0;                  // index bits the pass sweeps
  int64_t y_off = 0, src_off = 0; // partner passes: first local row / first partner amplitude
};

int rdm_release_scratch();   // frees the cached scratch of dnm_reduced_density_matrix

}  // namespace dnm

struct dnm_mat {
  // operator as given (deep copies, BuildContext semantics)
  std::vector<int64_t> masks, mask_offsets, signs;
  std::vector<double> real_coeffs;
  dnm::SubOwned left, right;
  int64_t M = 0, N = 0, m_local = 0, n_local = 0;
  int64_t row0 = 0;               // first row / column this rank owns
  bool sc_pair = false;           // SpinConserve(L,k) on both sides: incremental-rank kernel
  // SpinConserve(L,k) on both sides with vectors in the internal layout (sc3.h): m_local / n_local then count the
  // padding too (they are what the vector kernels sweep), rows_local the rows
  bool use_sc3 = false;
  std::unique_ptr<dnm::Sc3Mat> sc3;
  int64_t rows_local = 0;
  int64_t win_min = 0, win_max = -1;   // partitioned SpinConserve: columns this rank reads (cached)
  int rank = 0, nranks = 1;
  int flags = 0;
  bool xparity = false;
  bool host_only = false;         // DNM_MAT_HOST_ONLY: plan and tables only, no device

  bool hypercube = false;        // Full/Full or Parity/Parity: index space is a hypercube
  // DNM_MAT_REAL_PACKED: vectors are real, two amplitudes per complex128 element (M, N, m_local, n_local count
  // elements; rows_local / row0 still describe the operator's rows for the norm kernel)
  bool real_packed = false;
  dnm::OpForm op;
  dnm::Plan plan;
  std::vector<std::unique_ptr<dnm::PassOnDevice>> local_passes, remote_passes;

  // generic-kernel tables
  dnm::DevBuf d_masks, d_offsets, d_signs, d_rcoeffs;
  dnm::DevMsc dmsc{};
  dnm::DevBuf d_pmasks, d_poffsets, d_psigns, d_prcoeffs;     // the same in the labelling of a relabelled sc3 layout
  dnm::DevMsc dmsc_sc3{};         // what the sc3 row kernel reads (dmsc unless the layout is relabelled)
  dnm::DevBuf d_sclow;            // SpinConserve kernel: 16-bit unranking table
  dnm::ScLow sclow{};
  dnm::DevBuf d_scblock;          // block kernel: lb-bit patterns grouped by popcount
  dnm::DevBuf d_scperm;           // block kernel: optional block order
  dnm::ScBlock scblock{};         // lb == 0: block kernel not used
  int sc_nfast = 0;               // masks that are chain bonds with local signs
  // evolve: the Krylov step size at which the driver last handed over to the Chebyshev expansion (0: never) and
  // the basis size it belonged to -- later calls on this operator skip the Krylov probe when the estimate still holds
  double expm_tstep = 0.0;
  int expm_m = 0;
  int expm_bound = 0;             // what the Lanczos probe of the norm bound found: +1 tight, -1 loose, 0 not probed
  dnm::DevBuf d_scmasks;          // SpinConserve kernel: per-mask precomputation (ScMask[nmasks])

  // transposed exchange (dnm_mat_set_exchange): the terms that flip no rank bit in the vectors' own layout, the
  // others with the rank bits swapped against the local field [tr_f, tr_f + p) -- both rank-local; owned by this handle
  dnm_mat *tr_lo = nullptr, *tr_hi = nullptr;
  int tr_f = -1;
  ~dnm_mat();

  dnm::DevBuf diag;              // cached diagonal (double[m_local]) if precomputed
  bool have_diag = false;
  double nrm = -1.0;             // ctx->nrm cache, -1 = unset (bpetsc_template_2.c:926-929)
  dnm::DevBuf scratch;
};
