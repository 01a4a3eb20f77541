// The partitioned multiply as ONE native call: y = A x on rank r of P with the rank exchange on RCCL
// (grouped ncclSend / ncclRecv over xGMI) running on the library's own second HIP stream under the rank-local
// kernels.  The reference handles its ranks inside C as well (MatMult_CPU_Fast / _General's MPI branches,
// src/dynamite/_backend/bpetsc_template_2.c:413-504, 787-879; the CUDA shell all-gathers x, bcuda_template_2.cu:161-171);
// here the exchange is what the operator needs and nothing more:
//   * Full / Parity on 2^p ranks: XOR-partner sub-blocks (dnm_mat_exchange_plan); the rank-local passes run while the
//     blocks travel, the partner passes follow;
//   * every other partition (SpinConserve, Explicit / Auto, projections, odd rank counts): the rank's column window is
//     assembled from the owners' blocks -- only the ranges its rows read -- while the part of the multiply that needs
//     nothing from other ranks runs (two tiled SpinConserve passes: the lo pass; otherwise: the rows that read only the
//     rank's own block), the rest follows.
//   * Full / Parity under the transposed exchange (dnm_mat_set_exchange; the default from four ranks on): ONE all-to-all
//     takes the state to the layout in which the rank bits are local, the terms that flip no rank bit run under it, the
//     others on the redistributed state -- sub-piece by sub-piece as the pieces land -- and the returning all-to-all's
//     pieces are added on arrival.
// A communicator can also stand for one rank of P inside ONE process ("loop-back": the peers' blocks live in the same
// device memory and every message is an RCCL send to the process itself) -- how the schedules run, transport included,
// on the one-GPU boxes the tests have.
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "../../include/dynamite_amd.h"
#include "dnm_common.h"
#include "mat.h"

using namespace dnm;

// RCCL is bound at run time (dlopen), on the first communicator: the library itself loads without it (CPU-only hosts,
// the symbol checks of tests/test_abi.py), and a process that already holds a copy -- PyTorch ships its own -- shares
// that copy instead of loading a second one.  DNM_RCCL_LIB names a specific file.
#include <dlfcn.h>
namespace {
struct Rccl {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
int rccl_load() {
  if (g_rccl.h) return 0;
  void *h = nullptr;
  if (const char *e = getenv("DNM_RCCL_LIB")) h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
  for (const char *n : {"librccl.so", "librccl.so.1"})
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);           // a copy the process already has
  for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  DNM_CHECK(h, "RCCL not found (librccl.so.1; DNM_RCCL_LIB names a file): %s", dlerror());
  Rccl r;
  r.h = h;
#define DNM_SYM(name) DNM_CHECK((r.name = (decltype(r.name))dlsym(h, "nccl" #name)) != nullptr, "RCCL: no symbol nccl" #name)
  DNM_SYM(GetUniqueId); DNM_SYM(CommInitRank); DNM_SYM(CommDestroy); DNM_SYM(Send); DNM_SYM(Recv); DNM_SYM(GroupStart);
  DNM_SYM(GroupEnd); DNM_SYM(AllReduce); DNM_SYM(AllGather); DNM_SYM(GetErrorString);
#undef DNM_SYM
  g_rccl = r;
  return 0;
}
}  // namespace
#define ncclGetUniqueId g_rccl.GetUniqueId
#define ncclCommInitRank g_rccl.CommInitRank
#define ncclCommDestroy g_rccl.CommDestroy
#define ncclSend g_rccl.Send
#define ncclRecv g_rccl.Recv
#define ncclGroupStart g_rccl.GroupStart
#define ncclGroupEnd g_rccl.GroupEnd
#define ncclAllReduce g_rccl.AllReduce
#define ncclAllGather g_rccl.AllGather
#define ncclGetErrorString g_rccl.GetErrorString

#define DNM_NCCL(call)                                                                        \
  do {                                                                                        \
    ncclResult_t r_ = (call);                                                                 \
    if (r_ != ncclSuccess) {                                                                  \
      set_error("RCCL: %s (%s:%d)", ncclGetErrorString(r_), __FILE__, __LINE__);              \
      return 1;                                                                               \
    }                                                                                         \
  } while (0)

namespace {

struct Range { int64_t lo, hi; };        // [lo, hi)

// per-operator state of the window schedule
struct WindowState {
  bool ready = false;
  std::vector<int64_t> own0, ownn;               // every rank's block (positions of the right vector's own order)
  std::vector<int64_t> wlo, whi;                 // every rank's inclusive column window
  std::vector<std::vector<Range>> needs;         // ... and the ranges of it the rank reads
  DevBuf window;                                 // this rank's assembled window
  DevBuf xnat;                                   // this rank's block in index order (right vectors that are stored swizzled)
  int64_t wlen = 0;
  int split = 0;                                 // dnm_mat_window_split
  std::vector<Range> rows_local, rows_remote;    // row split of the other window kernels
};

struct PartnerState {
  std::vector<dnm_xfer> sends, recvs;
  std::vector<std::unique_ptr<DevBuf>> bufs;
  bool ready = false;
};

// per-operator state of the transposed exchange
struct TransposeState {
  bool ready = false;
  int p = 0, n = 0, f = 0, nb = 2;               // rank bits, local index bits, first bit of the exchanged field, pieces per peer
  int sub = 1;                                   // parts a piece travels in
  bool pipe = false;                             // the second part runs part by part as the pieces land
  int64_t cnt = 0, part = 0;                     // elements of a piece / of a part
  DevBuf xb, wb;                                 // the state in layout B; the second part's result there
  std::vector<hipEvent_t> ev;
  // loop-back: what the peers' second parts computed (their returning pieces), and the buffer their input is assembled in
  std::vector<std::unique_ptr<DevBuf>> peer_wb;
  DevBuf peer_xb;
  TransposeState() = default;
  TransposeState(const TransposeState &) = delete;
  ~TransposeState() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
  int event(size_t i, hipEvent_t *out) {
    while (ev.size() <= i) {
      hipEvent_t e;
      DNM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ev.push_back(e);
    }
    *out = ev[i];
    return 0;
  }
};

}  // namespace

struct dnm_comm {
  ncclComm_t nccl = nullptr;
  int rank = 0, nranks = 1;                      // of the RCCL communicator
  hipStream_t xs = nullptr;                      // exchange stream
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  DevBuf red;                                    // staging for reductions / gathers
  // loop-back: this process stands for rank vrank of vranks; peer q's block of x is peer_x[q], its handle peer_mat[q]
  int vrank = -1, vranks = 0;
  std::vector<const void *> peer_x;
  std::vector<dnm_mat *> peer_mat;
  std::map<dnm_mat *, WindowState> win;
  std::map<dnm_mat *, PartnerState> par;
  std::map<dnm_mat *, TransposeState> tr;
  int phase = DNM_PHASE_ALL;                     // dnm_comm_set_phase: the whole multiply, its messages alone, its kernels alone
  bool msgs() const { return phase != DNM_PHASE_COMPUTE; }
  bool kernels() const { return phase != DNM_PHASE_EXCHANGE; }
  int me() const { return vrank >= 0 ? vrank : rank; }
  int world() const { return vrank >= 0 ? vranks : nranks; }
};

namespace {

constexpr int WINDOW_CHUNKS = 1024;              // resolution of the needed-columns map (backend.py: the same)

// Post one received block: `count` complex128 elements that rank q holds at `src_off` of ITS vector land in dst.
// Real ranks: a receive from q (q posts the matching send from its own list).  Loop-back: a send to this process from
// the peer's block, and its receive.
// (loop_src: the peer's vector the block comes from in loop-back when it is not its block of x)
int post_recv(dnm_comm *c, int q, int64_t src_off, int64_t count, void *dst, const void *loop_src = nullptr) {
  if (c->vrank >= 0) {
    DNM_CHECK(q >= 0 && q < c->vranks && (loop_src || c->peer_x[(size_t)q]), "loop-back: no block for rank %d", q);
    const void *src = loop_src ? loop_src : c->peer_x[(size_t)q];
    DNM_NCCL(ncclSend((const char *)src + src_off * 16, (size_t)count * 2, ncclDouble, 0, c->nccl, c->xs));
    DNM_NCCL(ncclRecv(dst, (size_t)count * 2, ncclDouble, 0, c->nccl, c->xs));
    return 0;
  }
  DNM_NCCL(ncclRecv(dst, (size_t)count * 2, ncclDouble, q, c->nccl, c->xs));
  return 0;
}
int post_send(dnm_comm *c, int q, const void *x, int64_t off, int64_t count) {
  if (c->vrank >= 0) return 0;                   // the peers of a loop-back communicator receive nothing
  DNM_NCCL(ncclSend((const char *)x + off * 16, (size_t)count * 2, ncclDouble, q, c->nccl, c->xs));
  return 0;
}

// ranges covering the marked chunks of a window (backend.needed_ranges)
std::vector<Range> ranges_of(const std::vector<uint8_t> &map, int shift, int64_t wlo, int64_t whi_excl) {
  std::vector<Range> out;
  const int64_t first = wlo >> shift;
  size_t i = 0;
  while (i < map.size()) {
    if (!map[i]) { ++i; continue; }
    size_t j = i;
    while (j + 1 < map.size() && map[j + 1]) ++j;
    out.push_back({std::max(wlo, (first + (int64_t)i) << shift), std::min(whi_excl, (first + (int64_t)j + 1) << shift)});
    i = j + 1;
  }
  return out;
}

// one rank's window and needs from its handle (device sweeps, cached in the handle)
int window_of(dnm_mat *A, int64_t *lo, int64_t *hi, std::vector<Range> *needs, hipStream_t st) {
  DNM_TRY(dnm_mat_column_window(A, lo, hi, st));
  {
    // exact runs of needed blocks where the library has them (SpinConserve in the internal layout), as long as they fit
    // the fixed-size record the ranks exchange; else the chunk map
    int64_t n = 0;
    DNM_TRY(dnm_mat_column_ranges(A, 0, nullptr, &n));
    if (n > 0 && n <= WINDOW_CHUNKS) {
      std::vector<int64_t> rg((size_t)(2 * n));
      DNM_TRY(dnm_mat_column_ranges(A, n, rg.data(), &n));
      needs->clear();
      for (int64_t i = 0; i < n; ++i) needs->push_back({rg[(size_t)(2 * i)], rg[(size_t)(2 * i + 1)]});
      return 0;
    }
  }
  int shift = 0;
  while (((*hi - *lo + 1) >> shift) > WINDOW_CHUNKS) ++shift;
  const int64_t n = (*hi >> shift) - (*lo >> shift) + 1;
  std::vector<uint8_t> map((size_t)n, 0);
  DNM_TRY(dnm_mat_column_chunks(A, shift, map.data(), n, st));
  *needs = ranges_of(map, shift, *lo, *hi + 1);
  return 0;
}

// the block of the right vector rank q owns, in the order the window is expressed in
int ownership_of(const dnm_mat *A, int q, int P, int64_t *o0, int64_t *on) {
  if (A->use_sc3) {
    const std::vector<uint32_t> Tb = sc3_partition(*A->sc3->ly, P);
    int64_t is, il, ns, nl;
    sc3_range(*A->sc3->ly, Tb[(size_t)q], Tb[(size_t)q + 1], &is, &il, &ns, &nl);
    *o0 = A->real_packed ? is / 2 : is;
    *on = A->real_packed ? il / 2 : il;
    return 0;
  }
  const int64_t qn = A->N / P, rem = A->N % P;
  *o0 = (int64_t)q * qn + std::min<int64_t>(q, rem);
  *on = qn + (q < rem ? 1 : 0);
  return 0;
}

int setup_windows(dnm_comm *c, dnm_mat *A, WindowState &W, hipStream_t st) {
  const int P = c->world(), me = c->me();
  W.own0.assign((size_t)P, 0); W.ownn.assign((size_t)P, 0);
  W.wlo.assign((size_t)P, 0); W.whi.assign((size_t)P, -1);
  W.needs.assign((size_t)P, {});
  for (int q = 0; q < P; ++q) DNM_TRY(ownership_of(A, q, P, &W.own0[(size_t)q], &W.ownn[(size_t)q]));
  DNM_TRY(window_of(A, &W.wlo[(size_t)me], &W.whi[(size_t)me], &W.needs[(size_t)me], st));
  if (c->vrank >= 0) {
    // loop-back: the peers' windows from their handles (what they need decides nothing this rank does: it sends nothing)
    for (int q = 0; q < P; ++q)
      if (q != me && c->peer_mat[(size_t)q])
        DNM_TRY(window_of(c->peer_mat[(size_t)q], &W.wlo[(size_t)q], &W.whi[(size_t)q], &W.needs[(size_t)q], st));
  } else {
    // every rank's (window, needed ranges) to every rank: fixed-size records through one all-gather
    constexpr int MAXR = 2 * WINDOW_CHUNKS + 4;
    const size_t rec = (size_t)MAXR;
    std::vector<int64_t> mine(rec, 0), all(rec * (size_t)P, 0);
    mine[0] = W.wlo[(size_t)me]; mine[1] = W.whi[(size_t)me]; mine[2] = (int64_t)W.needs[(size_t)me].size();
    DNM_CHECK(2 * W.needs[(size_t)me].size() + 3 <= rec, "internal: too many window ranges");
    for (size_t i = 0; i < W.needs[(size_t)me].size(); ++i) {
      mine[3 + 2 * i] = W.needs[(size_t)me][i].lo;
      mine[4 + 2 * i] = W.needs[(size_t)me][i].hi;
    }
    if (c->red.bytes < rec * 8 * (size_t)(P + 1)) DNM_TRY(c->red.alloc(rec * 8 * (size_t)(P + 1)));
    char *dsend = (char *)c->red.p, *drecv = dsend + rec * 8;
    DNM_HIP(hipMemcpyAsync(dsend, mine.data(), rec * 8, hipMemcpyHostToDevice, c->xs));
    DNM_NCCL(ncclAllGather(dsend, drecv, rec, ncclInt64, c->nccl, c->xs));
    DNM_HIP(hipMemcpyAsync(all.data(), drecv, rec * 8 * (size_t)P, hipMemcpyDeviceToHost, c->xs));
    DNM_HIP(hipStreamSynchronize(c->xs));
    for (int q = 0; q < P; ++q) {
      const int64_t *r = all.data() + rec * (size_t)q;
      W.wlo[(size_t)q] = r[0]; W.whi[(size_t)q] = r[1];
      W.needs[(size_t)q].clear();
      for (int64_t i = 0; i < r[2]; ++i) W.needs[(size_t)q].push_back({r[3 + 2 * i], r[4 + 2 * i]});
    }
  }
  W.wlen = W.whi[(size_t)me] - W.wlo[(size_t)me] + 1;
  DNM_CHECK(W.wlen >= 1, "internal: empty column window");
  DNM_TRY(W.window.alloc((size_t)W.wlen * 16));
  DNM_HIP(hipMemsetAsync(W.window.p, 0, (size_t)W.wlen * 16, st));       // what never travels is never read, but is defined
  DNM_TRY(dnm_mat_window_split(A, &W.split));
  W.rows_local.clear(); W.rows_remote.clear();
  if (!W.split) {
    // rows that read only the rank's own block of x run under the exchange (backend.ShellMat._window_row_ranges)
    constexpr int MAXRG = 8;
    int64_t buf[2 * MAXRG];
    int n = 0;
    DNM_TRY(dnm_mat_window_local_rows(A, W.own0[(size_t)me], W.own0[(size_t)me] + W.ownn[(size_t)me], MAXRG, 64, buf, &n, st));
    int64_t covered = 0;
    for (int i = 0; i < n; ++i) covered += buf[2 * i + 1] - buf[2 * i];
    if (n > 0 && covered * 20 >= A->m_local) {
      int64_t at = 0;
      for (int i = 0; i < n; ++i) {
        W.rows_local.push_back({buf[2 * i], buf[2 * i + 1]});
        if (buf[2 * i] > at) W.rows_remote.push_back({at, buf[2 * i]});
        at = buf[2 * i + 1];
      }
      if (at < A->m_local) W.rows_remote.push_back({at, A->m_local});
    }
  }
  W.ready = true;
  return 0;
}

int mult_window(dnm_comm *c, dnm_mat *A, const void *x, void *y, hipStream_t st) {
  WindowState &W = c->win[A];
  if (!W.ready) DNM_TRY(setup_windows(c, A, W, st));
  const int P = c->world(), me = c->me();
  const int64_t wlo = W.wlo[(size_t)me], whi = W.whi[(size_t)me] + 1;
  const int64_t my0 = W.own0[(size_t)me], myn = W.ownn[(size_t)me];
  // The window is expressed in index order (or, for SpinConserve vectors in the internal layout, in positions of the
  // layout, where blocks travel as they lie): a block of a Full / Parity vector that is stored XOR-swizzled (odd rank counts,
  // projections between subspaces) is straightened first -- what the peers receive and what the rank's own part of the
  // window holds is index order.
  if (!A->use_sc3 && A->right.host.swz != 0) {
    DNM_CHECK(c->vrank < 0, "loop-back: window partitions of swizzled vectors are not looped back");
    if (W.xnat.bytes < (size_t)A->n_local * 16) DNM_TRY(W.xnat.alloc((size_t)A->n_local * 16));
    DNM_TRY(dnm_vec_swizzle_copy(W.xnat.p, x, A->n_local, A->right.host.swz, st));
    x = W.xnat.p;
  }
  // the exchange: x is ready when the compute stream gets here
  DNM_HIP(hipEventRecord(c->ev_ready, st));
  DNM_HIP(hipStreamWaitEvent(c->xs, c->ev_ready, 0));
  if (c->msgs()) DNM_NCCL(ncclGroupStart());
  for (int q = 0; q < P && c->msgs(); ++q) {
    if (q == me) continue;
    const int64_t q0 = W.own0[(size_t)q], qn = W.ownn[(size_t)q];
    for (const Range &r : W.needs[(size_t)me]) {           // what this rank reads of q's block
      const int64_t lo = std::max(r.lo, q0), hi = std::min(r.hi, q0 + qn);
      if (lo < hi) DNM_TRY(post_recv(c, q, lo - q0, hi - lo, (char *)W.window.p + (lo - wlo) * 16));
    }
    for (const Range &r : W.needs[(size_t)q]) {            // what q reads of this rank's block
      const int64_t lo = std::max(r.lo, my0), hi = std::min(r.hi, my0 + myn);
      if (lo < hi) DNM_TRY(post_send(c, q, x, lo - my0, hi - lo));
    }
  }
  if (c->msgs()) DNM_NCCL(ncclGroupEnd());
  DNM_HIP(hipEventRecord(c->ev_done, c->xs));
  if (!c->kernels()) {                           // the messages alone (dnm_comm_set_phase): the caller's stream sees them done
    DNM_HIP(hipStreamWaitEvent(st, c->ev_done, 0));
    return 0;
  }
  // the rank's own part of its window
  const int64_t a = std::max(wlo, my0), b = std::min(whi, my0 + myn);
  if (a < b) DNM_TRY(dnm_vec_copy((const char *)x + (a - my0) * 16, (char *)W.window.p + (a - wlo) * 16, b - a, st));
  if (W.split) {
    DNM_TRY(dnm_mat_mult_window_local(A, x, y, st));                  // under the exchange
    DNM_HIP(hipStreamWaitEvent(st, c->ev_done, 0));
    return dnm_mat_mult_window_remote(A, W.window.p, wlo, W.wlen, y, st);
  }
  if (!W.rows_local.empty()) {
    for (const Range &r : W.rows_local) DNM_TRY(dnm_mat_mult_window_rows(A, W.window.p, wlo, W.wlen, y, r.lo, r.hi, st));
    DNM_HIP(hipStreamWaitEvent(st, c->ev_done, 0));
    for (const Range &r : W.rows_remote) DNM_TRY(dnm_mat_mult_window_rows(A, W.window.p, wlo, W.wlen, y, r.lo, r.hi, st));
    return 0;
  }
  DNM_HIP(hipStreamWaitEvent(st, c->ev_done, 0));
  return dnm_mat_mult_window(A, W.window.p, wlo, W.wlen, y, st);
}

int setup_partner(dnm_mat *A, PartnerState &Q) {
  {
    int ns = 0, nr = 0;
    DNM_TRY(dnm_mat_exchange_plan(A, &ns, nullptr, &nr, nullptr));
    Q.sends.resize((size_t)std::max(1, ns));
    Q.recvs.resize((size_t)std::max(1, nr));
    DNM_TRY(dnm_mat_exchange_plan(A, &ns, Q.sends.data(), &nr, Q.recvs.data()));
    Q.sends.resize((size_t)ns);
    Q.recvs.resize((size_t)nr);
    for (const dnm_xfer &r : Q.recvs) {
      Q.bufs.emplace_back(new DevBuf());
      DNM_TRY(Q.bufs.back()->alloc((size_t)r.count * 16));
    }
    Q.ready = true;
  }
  return 0;
}

int mult_partner(dnm_comm *c, dnm_mat *A, const void *x, void *y, hipStream_t st) {
  PartnerState &Q = c->par[A];
  if (!Q.ready) DNM_TRY(setup_partner(A, Q));
  if (Q.recvs.empty() && Q.sends.empty()) return dnm_mat_mult(A, x, y, st);
  DNM_HIP(hipEventRecord(c->ev_ready, st));
  DNM_HIP(hipStreamWaitEvent(c->xs, c->ev_ready, 0));
  if (c->msgs()) {
    DNM_NCCL(ncclGroupStart());
    for (const dnm_xfer &s : Q.sends) DNM_TRY(post_send(c, s.partner, x, s.offset, s.count));
    for (size_t i = 0; i < Q.recvs.size(); ++i)
      DNM_TRY(post_recv(c, Q.recvs[i].partner, Q.recvs[i].offset, Q.recvs[i].count, Q.bufs[i]->p));
    DNM_NCCL(ncclGroupEnd());
  }
  DNM_HIP(hipEventRecord(c->ev_done, c->xs));
  if (c->kernels()) DNM_TRY(dnm_mat_mult_local(A, x, y, st));         // under the exchange
  DNM_HIP(hipStreamWaitEvent(st, c->ev_done, 0));
  for (size_t i = 0; i < Q.recvs.size() && c->kernels(); ++i) DNM_TRY(dnm_mat_mult_remote(A, (int32_t)i, Q.bufs[i]->p, y, st));
  return 0;
}

constexpr int TR_SUB = 4;                        // parts a piece of the transposed exchange travels in (backend.ShellMat.TR_SUB)

int setup_transposed(dnm_comm *c, dnm_mat *A, TransposeState &T) {
  const int P = c->world();
  T.p = 0;
  while ((1 << T.p) < P) ++T.p;
  T.n = 0;
  while (((int64_t)1 << T.n) < A->n_local) ++T.n;
  DNM_CHECK(((int64_t)1 << T.n) == A->n_local && (1 << T.p) == P, "internal: transposed exchange on blocks that are not subcubes");
  T.f = A->tr_f;
  T.cnt = (int64_t)1 << T.f;
  T.nb = 1 << (T.n - T.f - T.p);
  // sub-piece pipelining of the second part: its ranges of workgroups must be the top bits inside a piece and must
  // not read outside themselves (backend.ShellMat.set_transposed: the same conditions)
  int top = -1, gathers = 0;
  DNM_TRY(dnm_mat_local_part_bits(A->tr_hi, &top, &gathers));
  int logsub = 0;
  while ((1 << logsub) < TR_SUB) ++logsub;
  const int swz = A->right.host.swz;
  const char *pk = knob("DNM_TRANSPOSE_PIPE");
  T.pipe = !(pk && pk[0] == '0') && top == T.f - 1 && gathers == 0 && T.cnt % TR_SUB == 0 &&
           T.n - A->tr_hi->plan.cfg.B >= logsub && (swz == 0 || 2 * swz - 4 <= T.f - logsub);
  T.sub = T.pipe ? TR_SUB : ((T.cnt % TR_SUB == 0 && T.cnt / TR_SUB >= 1024) ? TR_SUB : 1);
  T.part = T.cnt / T.sub;
  DNM_TRY(T.xb.alloc((size_t)A->n_local * 16));
  DNM_TRY(T.wb.alloc((size_t)A->n_local * 16));
  T.ready = true;
  return 0;
}

// y = A x with the transposed exchange (see the header of this file and dnm_mat_set_exchange)
int mult_transposed(dnm_comm *c, dnm_mat *A, const void *x, void *y, hipStream_t st) {
  TransposeState &T = c->tr[A];
  if (!T.ready) DNM_TRY(setup_transposed(c, A, T));
  const int P = c->world(), me = c->me();
  const int64_t cnt = T.cnt, part = T.part;
  const int sub = T.sub;
  const bool ker = c->kernels();                 // (dnm_comm_set_phase: the messages alone skip every kernel)
  auto off = [&](int b, int q) { return ((int64_t)b * P + q) << T.f; };
  auto at = [](const void *v, int64_t o) { return (void *)((const char *)v + o * 16); };
  // loop-back with the peers' handles: what comes back is what THEY computed -- run their second parts on the state
  // as it would reach them (validation at test sizes; without handles the returning pieces carry this rank's own
  // result: the traffic of the schedule, for timing)
  std::vector<const void *> back_src((size_t)P, nullptr);
  if (c->vrank >= 0) {
    bool have = true;
    for (int q = 0; q < P; ++q) have = have && (q == me || (c->peer_mat[(size_t)q] && c->peer_mat[(size_t)q]->tr_hi && c->peer_x[(size_t)q]));
    if (have && ker) {
      if (T.peer_wb.size() != (size_t)P) {
        T.peer_wb.clear();
        for (int q = 0; q < P; ++q) {
          T.peer_wb.emplace_back(new DevBuf());
          if (q != me) DNM_TRY(T.peer_wb.back()->alloc((size_t)A->n_local * 16));
        }
        DNM_TRY(T.peer_xb.alloc((size_t)A->n_local * 16));
      }
      for (int q = 0; q < P; ++q) {
        if (q == me) continue;
        for (int r = 0; r < P; ++r)
          for (int b = 0; b < T.nb; ++b)
            DNM_TRY(dnm_vec_copy(at(r == me ? x : c->peer_x[(size_t)r], off(b, q)), at(T.peer_xb.p, off(b, r)), cnt, st));
        DNM_TRY(dnm_mat_mult_local(c->peer_mat[(size_t)q]->tr_hi, T.peer_xb.p, T.peer_wb[(size_t)q]->p, st));
        back_src[(size_t)q] = T.peer_wb[(size_t)q]->p;
      }
    } else {
      for (int q = 0; q < P; ++q) back_src[(size_t)q] = T.wb.p;
    }
  }
  // one group of the all-to-all: elements [o, o + len) of the pieces with index b in [b0, b1) -- `src` goes out,
  // the peers' land in `dst` at the same offsets (the map between the layouts is its own inverse)
  auto post = [&](const void *src, void *dst, int b0, int b1, int64_t o, int64_t len, bool returning) -> int {
    if (!c->msgs()) return 0;
    DNM_NCCL(ncclGroupStart());
    for (int q = 0; q < P; ++q) {
      if (q == me) continue;
      for (int b = b0; b < b1; ++b) {
        DNM_TRY(post_send(c, q, src, off(b, q) + o, len));
        DNM_TRY(post_recv(c, q, off(b, me) + o, len, at(dst, off(b, q) + o), returning ? back_src[(size_t)q] : nullptr));
      }
    }
    DNM_NCCL(ncclGroupEnd());
    return 0;
  };
  auto add = [&](const void *src, int64_t o, int64_t len) { return ker ? dnm_vec_axpby(at(y, o), at(src, o), len, 1.0, 0.0, 1.0, 0.0, st) : 0; };
  size_t nev = 0;
  hipEvent_t e;
  // x is ready when the compute stream gets here
  DNM_HIP(hipEventRecord(c->ev_ready, st));
  DNM_HIP(hipStreamWaitEvent(c->xs, c->ev_ready, 0));
  if (T.pipe) {
    // part s of ALL pieces is what range s of the second part's workgroups reads and writes
    std::vector<hipEvent_t> fwd((size_t)sub), back((size_t)sub);
    for (int s = 0; s < sub; ++s) {
      DNM_TRY(post(x, T.xb.p, 0, T.nb, s * part, part, false));
      DNM_TRY(T.event(nev++, &fwd[(size_t)s]));
      DNM_HIP(hipEventRecord(fwd[(size_t)s], c->xs));
    }
    for (int b = 0; b < T.nb && ker; ++b) DNM_TRY(dnm_vec_copy(at(x, off(b, me)), at(T.xb.p, off(b, me)), cnt, st));
    if (ker) DNM_TRY(dnm_mat_mult_local(A->tr_lo, x, y, st));           // under the all-to-all
    for (int s = 0; s < sub; ++s) {
      DNM_HIP(hipStreamWaitEvent(st, fwd[(size_t)s], 0));
      if (ker) DNM_TRY(dnm_mat_mult_local_part(A->tr_hi, T.xb.p, T.wb.p, s, sub, st));
      // part s of xb has been consumed: it takes the returning part s
      DNM_TRY(T.event(nev++, &e));
      DNM_HIP(hipEventRecord(e, st));
      DNM_HIP(hipStreamWaitEvent(c->xs, e, 0));
      DNM_TRY(post(T.wb.p, T.xb.p, 0, T.nb, s * part, part, true));
      DNM_TRY(T.event(nev++, &back[(size_t)s]));
      DNM_HIP(hipEventRecord(back[(size_t)s], c->xs));
      for (int b = 0; b < T.nb; ++b) DNM_TRY(add(T.wb.p, off(b, me) + s * part, part));
    }
    for (int s = 0; s < sub; ++s) {
      DNM_HIP(hipStreamWaitEvent(st, back[(size_t)s], 0));
      for (int q = 0; q < P; ++q)
        for (int b = 0; b < T.nb && q != me; ++b) DNM_TRY(add(T.xb.p, off(b, q) + s * part, part));
    }
    return 0;
  }
  DNM_TRY(post(x, T.xb.p, 0, T.nb, 0, cnt, false));
  DNM_TRY(T.event(nev++, &e));
  DNM_HIP(hipEventRecord(e, c->xs));
  for (int b = 0; b < T.nb && ker; ++b) DNM_TRY(dnm_vec_copy(at(x, off(b, me)), at(T.xb.p, off(b, me)), cnt, st));
  if (ker) DNM_TRY(dnm_mat_mult_local(A->tr_lo, x, y, st));             // under the all-to-all
  DNM_HIP(hipStreamWaitEvent(st, e, 0));
  if (ker) DNM_TRY(dnm_mat_mult_local(A->tr_hi, T.xb.p, T.wb.p, st));
  // the way back (xb is free again) in nb * sub batches -- every piece travels as `sub` contiguous parts, part by part
  // over all peers: what a batch brought is added to y while the next ones are on the links
  DNM_TRY(T.event(nev++, &e));
  DNM_HIP(hipEventRecord(e, st));
  DNM_HIP(hipStreamWaitEvent(c->xs, e, 0));
  std::vector<hipEvent_t> got((size_t)T.nb * (size_t)sub);
  for (int b = 0; b < T.nb; ++b)
    for (int k = 0; k < sub; ++k) {
      DNM_TRY(post(T.wb.p, T.xb.p, b, b + 1, k * part, part, true));
      DNM_TRY(T.event(nev++, &got[(size_t)b * sub + k]));
      DNM_HIP(hipEventRecord(got[(size_t)b * sub + k], c->xs));
    }
  for (int b = 0; b < T.nb; ++b) DNM_TRY(add(T.wb.p, off(b, me), cnt));
  for (int b = 0; b < T.nb; ++b)
    for (int k = 0; k < sub; ++k) {
      DNM_HIP(hipStreamWaitEvent(st, got[(size_t)b * sub + k], 0));
      for (int q = 0; q < P; ++q)
        if (q != me) DNM_TRY(add(T.xb.p, off(b, q) + k * part, part));
    }
  return 0;
}

struct HookCtx {
  dnm_comm *c;
  dnm_mat *A;
  hipStream_t st;
};
std::vector<std::unique_ptr<HookCtx>> g_hookctx;

int hook_mult(void *ctx, const void *x, void *y) {
  HookCtx *h = (HookCtx *)ctx;
  return dnm_mat_mult_partitioned(h->A, h->c, x, y, h->st);
}
int hook_sum(void *ctx, double *buf, int n) { return dnm_comm_allreduce(((HookCtx *)ctx)->c, buf, n, 0); }
int hook_max(void *ctx, double *buf, int n) { return dnm_comm_allreduce(((HookCtx *)ctx)->c, buf, n, 1); }

}  // namespace

extern "C" {

int dnm_comm_unique_id(void *id128) {
  DNM_CHECK(id128, "null argument");
  DNM_TRY(rccl_load());
  ncclUniqueId id;
  DNM_NCCL(ncclGetUniqueId(&id));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, sizeof id);
  return 0;
}

int dnm_comm_create(const void *id128, int rank, int nranks, dnm_comm **out) {
  DNM_CHECK(id128 && out && nranks >= 1 && rank >= 0 && rank < nranks, "bad argument");
  *out = nullptr;
  DNM_TRY(rccl_load());
  std::unique_ptr<dnm_comm> c(new dnm_comm());
  c->rank = rank;
  c->nranks = nranks;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  DNM_NCCL(ncclCommInitRank(&c->nccl, nranks, id, rank));
  // The exchange stream gets the HIGHEST priority the device offers.  Streams of one priority share the process's few
  // hardware queues round-robin (GPU_MAX_HW_QUEUES, default 4): on this system an RCCL transfer on one stream and a kernel on
  // another ran ONE AFTER THE OTHER whenever the two streams landed on the same queue (tools/probes/rccl_concurrency_probe.py:
  // 25.3 ms together = 3.7 + 20.6, against 21.4 with eight queues) -- no overlap of exchange and compute at all.  A
  // high-priority stream has a queue of its own class, and RCCL's few workgroups get their slots ahead of the bulk of the
  // rank-local pass's.  DNM_COMM_PRIORITY=0: the default priority (A/B runs).
  {
    int least = 0, greatest = 0;
    DNM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const char *pe = getenv("DNM_COMM_PRIORITY");
    if (pe && pe[0] == '0') DNM_HIP(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
    else DNM_HIP(hipStreamCreateWithPriority(&c->xs, hipStreamNonBlocking, greatest));
  }
  DNM_HIP(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  DNM_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  DNM_TRY(c->red.alloc(1 << 16));
  *out = c.release();
  return 0;
}

int dnm_comm_destroy(dnm_comm *c) {
  if (!c) return 0;
  for (auto it = g_hookctx.begin(); it != g_hookctx.end();)
    it = ((*it)->c == c) ? g_hookctx.erase(it) : it + 1;
  if (c->xs) (void)hipStreamSynchronize(c->xs);
  if (c->nccl) (void)ncclCommDestroy(c->nccl);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->xs) (void)hipStreamDestroy(c->xs);
  delete c;
  return 0;
}

int dnm_comm_forget(dnm_comm *c, dnm_mat *A) {
  DNM_CHECK(c, "null communicator");
  if (c->xs) DNM_HIP(hipStreamSynchronize(c->xs));      // buffers and events of a schedule still running go with it
  c->win.erase(A);
  c->par.erase(A);
  c->tr.erase(A);
  return 0;
}

int dnm_comm_loopback(dnm_comm *c, int vrank, int vranks, const void *const *peer_x, dnm_mat *const *peer_mat) {
  DNM_CHECK(c && c->nranks == 1, "loop-back needs a communicator of one rank");
  DNM_CHECK(vranks >= 1 && vrank >= 0 && vrank < vranks && peer_x, "bad argument");
  c->vrank = vrank;
  c->vranks = vranks;
  c->peer_x.assign(peer_x, peer_x + vranks);
  c->peer_mat.assign((size_t)vranks, nullptr);
  if (peer_mat) c->peer_mat.assign(peer_mat, peer_mat + vranks);
  c->win.clear();
  c->par.clear();
  c->tr.clear();
  return 0;
}

int dnm_comm_allreduce(dnm_comm *c, double *vals, int n, int op) {
  DNM_CHECK(c && (vals || n == 0) && n >= 0 && (op == 0 || op == 1), "bad argument");
  if (n == 0 || (c->nranks == 1 && c->vrank < 0)) return 0;
  if (c->vrank >= 0) return 0;                   // loop-back: one process, nothing to add
  if (c->red.bytes < (size_t)n * 8) DNM_TRY(c->red.alloc((size_t)n * 8));
  DNM_HIP(hipMemcpyAsync(c->red.p, vals, (size_t)n * 8, hipMemcpyHostToDevice, c->xs));
  DNM_NCCL(ncclAllReduce(c->red.p, c->red.p, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->nccl, c->xs));
  DNM_HIP(hipMemcpyAsync(vals, c->red.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->xs));
  DNM_HIP(hipStreamSynchronize(c->xs));
  return 0;
}

int dnm_mat_mult_partitioned(dnm_mat *A, dnm_comm *c, const void *x, void *y, void *stream) {
  DNM_CHECK(A && c && x && y && x != y, "bad argument");
  DNM_CHECK(!A->host_only, "host-only matrix");
  DNM_CHECK(A->nranks == c->world() && A->rank == c->me(), "the matrix is rank %d of %d, the communicator rank %d of %d",
            A->rank, A->nranks, c->me(), c->world());
  hipStream_t st = (hipStream_t)stream;
  if (A->nranks == 1) return dnm_mat_mult(A, x, y, stream);
  if (A->tr_hi) return mult_transposed(c, A, x, y, st);
  if (A->hypercube && A->plan.use_tiled) return mult_partner(c, A, x, y, st);
  return mult_window(c, A, x, y, st);
}

int dnm_comm_set_phase(dnm_comm *c, int phase) {
  DNM_CHECK(c && (phase == DNM_PHASE_ALL || phase == DNM_PHASE_EXCHANGE || phase == DNM_PHASE_COMPUTE), "bad argument");
  c->phase = phase;
  return 0;
}

int dnm_comm_prepare(dnm_comm *c, dnm_mat *A, void *stream) {
  DNM_CHECK(A && c, "null argument");
  DNM_CHECK(A->nranks == c->world() && A->rank == c->me(), "the matrix is rank %d of %d, the communicator rank %d of %d",
            A->rank, A->nranks, c->me(), c->world());
  if (A->nranks == 1 || A->host_only) return 0;
  if (A->tr_hi) {
    TransposeState &T = c->tr[A];
    return T.ready ? 0 : setup_transposed(c, A, T);
  }
  if (A->hypercube && A->plan.use_tiled) {
    PartnerState &Q = c->par[A];
    return Q.ready ? 0 : setup_partner(A, Q);
  }
  WindowState &W = c->win[A];
  return W.ready ? 0 : setup_windows(c, A, W, (hipStream_t)stream);
}

int dnm_comm_hooks(dnm_comm *c, dnm_mat *A, void *stream, dnm_hooks *out) {
  DNM_CHECK(c && A && out, "null argument");
  g_hookctx.emplace_back(new HookCtx{c, A, (hipStream_t)stream});
  out->ctx = g_hookctx.back().get();
  out->mult = hook_mult;
  out->allreduce_sum = hook_sum;
  out->allreduce_max = hook_max;
  return 0;
}

}  // extern "C"
