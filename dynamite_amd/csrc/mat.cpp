// C ABI: device helpers, subspace maps, and the shell matrix (create / mult /
// norm / diagonal / destroy).  See include/dynamite_amd.h for the reference
// interfaces each entry point replaces.
#include "mat.h"
#include "vec_api.h"

#include <algorithm>
#include <cstring>

namespace dnm {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

int DevBuf::alloc(size_t nbytes) {
  release();
  if (nbytes == 0) nbytes = 16;
  DNM_HIP(hipMalloc(&p, nbytes));
  bytes = nbytes;
  return 0;
}
int DevBuf::upload(const void *host, size_t nbytes) {
  DNM_TRY(alloc(nbytes));
  if (nbytes) DNM_HIP(hipMemcpy(p, host, nbytes, hipMemcpyHostToDevice));
  return 0;
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  bytes = 0;
}

static int view_from_c(const dnm_subspace *s, SubView *v) {
  DNM_CHECK(s != nullptr, "null subspace descriptor");
  DNM_CHECK(s->type >= DNM_FULL && s->type <= DNM_SPIN_CONSERVE, "unknown subspace type %d", s->type);
  DNM_CHECK(s->L >= 1 && s->L <= 63, "L=%lld out of range", (long long)s->L);
  v->type = s->type;
  v->L = (int32_t)s->L;
  v->space = (int32_t)s->space;
  v->k = (int32_t)s->k;
  v->ld = (int32_t)s->ld_nchoosek;
  v->dim = s->dim;
  v->nchoosek = s->nchoosek;
  v->state_map = s->state_map;
  v->rmap_indices = s->rmap_indices;
  v->rmap_states = s->rmap_states;
  v->swz = s->vec_swizzle;
  v->sc3 = 0;
  if (s->type == DNM_SPIN_CONSERVE && v->swz != 0) {       // the internal layout of sc3.h: a | w << 8
    v->sc3 = v->swz;
    v->swz = 0;
    DNM_CHECK(sc3_valid((int)s->L, (int)s->k, sc3_code_a(v->sc3), sc3_code_w(v->sc3)),
              "vec_swizzle %d: no such SpinConserve layout for L=%d k=%d", v->sc3, (int)s->L, (int)s->k);
  }
  DNM_CHECK(v->swz == 0 || ((s->type == DNM_FULL || s->type == DNM_PARITY) && v->swz >= 5 && v->swz <= 24),
            "vec_swizzle %d: swizzled vectors need a Full or Parity subspace and a shift in [5, 24]", v->swz);
  if (s->type == DNM_PARITY) DNM_CHECK(s->space == 0 || s->space == 1, "parity space must be 0 or 1");
  if (s->type == DNM_SPIN_CONSERVE) {
    DNM_CHECK(s->nchoosek != nullptr && s->ld_nchoosek == s->L + 1 && s->k >= 0 && s->k <= s->L,
              "bad SpinConserve descriptor");
  }
  if (s->type == DNM_EXPLICIT)
    DNM_CHECK(s->state_map && s->rmap_states && s->dim >= 1, "bad Explicit descriptor");
  return 0;
}

int SubOwned::init(const dnm_subspace *s, bool want_device) {
  SubView v{};
  DNM_TRY(view_from_c(s, &v));
  host = v;
  if (v.type == DNM_SPIN_CONSERVE) {
    nck.assign(v.nchoosek, v.nchoosek + (size_t)(v.k + 1) * v.ld);
    host.nchoosek = nck.data();
  }
  if (v.type == DNM_EXPLICIT) {
    smap.assign(v.state_map, v.state_map + v.dim);
    rstates.assign(v.rmap_states, v.rmap_states + v.dim);
    host.state_map = smap.data();
    host.rmap_states = rstates.data();
    if (v.rmap_indices) {
      rind.assign(v.rmap_indices, v.rmap_indices + v.dim);
      host.rmap_indices = rind.data();
    }
    // bucket table: about two buckets per state (at most 2^26: 512 MB), so a search touches one or two entries
    int tb = 1;
    int tb_max = 26;         // measured at 40 M states: 8.2 ms (2^20 buckets), 6.5 (2^22), 4.9 (2^24), 4.2 (2^26); binary search: 22.9
    if (const char *e = knob("DNM_BUCKET_BITS")) tb_max = atoi(e);
    while (tb < v.L && tb < tb_max && ((int64_t)1 << tb) < 2 * v.dim) ++tb;
    host.bucket_shift = v.L - tb;
    const int64_t nb = (int64_t)1 << tb;
    bucket.assign((size_t)nb + 1, 0);
    bool sorted = true;
    for (int64_t i = 0; i < v.dim; ++i) {
      const int64_t st = rstates[(size_t)i];
      if (st < 0 || (st >> host.bucket_shift) >= nb || (i > 0 && st <= rstates[(size_t)i - 1])) { sorted = false; break; }
      ++bucket[(size_t)(st >> host.bucket_shift) + 1];
    }
    if (sorted) {
      for (int64_t b = 0; b < nb; ++b) bucket[(size_t)b + 1] += bucket[(size_t)b];
      host.bucket = bucket.data();
    } else {          // not a sorted list of L-bit states: plain binary search over everything
      bucket.clear();
      host.bucket = nullptr;
    }
  }
  host.dim = sub_dim(host);
  dev = host;
  dev.nchoosek = nullptr;
  dev.state_map = dev.rmap_indices = dev.rmap_states = nullptr;
  dev.bucket = nullptr;
  if (want_device) {
    if (!nck.empty()) {
      DNM_TRY(d_nck.upload(nck.data(), nck.size() * 8));
      dev.nchoosek = (const int64_t *)d_nck.p;
    }
    if (!smap.empty()) {
      DNM_TRY(d_smap.upload(smap.data(), smap.size() * 8));
      DNM_TRY(d_rstates.upload(rstates.data(), rstates.size() * 8));
      dev.state_map = (const int64_t *)d_smap.p;
      dev.rmap_states = (const int64_t *)d_rstates.p;
      if (!rind.empty()) {
        DNM_TRY(d_rind.upload(rind.data(), rind.size() * 8));
        dev.rmap_indices = (const int64_t *)d_rind.p;
      }
      if (!bucket.empty()) {
        DNM_TRY(d_bucket.upload(bucket.data(), bucket.size() * 8));
        dev.bucket = (const int64_t *)d_bucket.p;
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------
// MSC -> row-evaluated index-space form (see plan.h).  Full: identity map.
// Parity: index = state >> 1; the dropped bit parity(idx)^space is folded
// into the sign masks (cf. the check_parity branch of sum_term,
// bpetsc_template_2.c:659-662, 848-854).
// ---------------------------------------------------------------------------
static int build_opform(const dnm_mat &A, OpForm *op) {
  const SubView &l = A.left.host, &r = A.right.host;
  const bool par = l.type == DNM_PARITY;
  const int n = par ? l.L - 1 : l.L;
  op->n = n;
  op->masks.clear();
  const uint64_t ones = n >= 64 ? ~0ull : (((uint64_t)1 << n) - 1);
  for (size_t mi = 0; mi < A.masks.size(); ++mi) {
    const uint64_t mask = (uint64_t)A.masks[mi];
    if (par && (parity64(mask) != (l.space ^ r.space))) continue;   // maps outside the right space
    RowMask rm;
    rm.mask = par ? (mask >> 1) : mask;
    for (int64_t t = A.mask_offsets[mi]; t < A.mask_offsets[mi + 1]; ++t) {
      const uint64_t sg = (uint64_t)A.signs[t];
      RowTerm rt;
      rt.is_imag = parity64(mask & sg);           // !TERM_REAL
      double c = A.real_coeffs[t];
      uint64_t s2 = sg;
      if (par) {
        s2 = sg >> 1;
        if (sg & 1) {
          s2 ^= ones;
          if (r.space) c = -c;
        }
      }
      // column-evaluated -> row-evaluated: col = row ^ mask
      if (parity64(rm.mask & s2)) c = -c;
      rt.sign = s2;
      rt.coeff = c;
      rm.terms.push_back(rt);
    }
    if (rm.mask == 0) {
      RowMask im;
      im.mask = 0;
      im.zero_mask_offdiag = true;
      std::vector<RowTerm> re;
      for (const RowTerm &t : rm.terms) (t.is_imag ? im.terms : re).push_back(t);
      rm.terms.swap(re);
      if (!im.terms.empty()) op->masks.push_back(std::move(im));
      if (rm.terms.empty()) continue;
    }
    op->masks.push_back(std::move(rm));
  }
  std::stable_sort(op->masks.begin(), op->masks.end(),
            [](const RowMask &a, const RowMask &b) { return a.mask < b.mask; });
  return 0;
}

static uint32_t compress_to_tile(uint64_t bits, const PassSpec &ps) {
  uint32_t out = 0;
  int off = 0;
  for (int j = 0; j < ps.nseg; ++j) {
    uint64_t seg = (bits >> ps.seg_pos[j]) & (((uint64_t)1 << ps.seg_len[j]) - 1);
    out |= (uint32_t)seg << off;
    off += ps.seg_len[j];
  }
  return out;
}

// Real-packed form of a real operator (DNM_MAT_REAL_PACKED): index r = 2 j + b, element j of a vector holds the
// amplitudes b = 0 (real part) and b = 1 (imaginary part).  A term (mask m, sign s, coefficient c) contributes
// c (-1)^popcount(r & s) x[r ^ m] to y[r]; with m' = m >> 1, s' = s >> 1, f = m & 1:
//   y[j].lane(b) += c (-1)^popcount(j & s') (-1)^(b (s & 1)) x[j ^ m'].lane(b ^ f)
// -- one coefficient per lane.  The form keeps the record layout: a term's is_imag names its lane, the records of a
// mask carry both lanes (slots 0, 1: lane 0; slots 2, 3: lane 1) and RowMask::pack_flip = f.  Diagonal terms whose
// sign reaches bit 0 differ between the lanes: they become a mask-0 off-diagonal entry (partner = the element itself).
static int pack_opform(OpForm *op) {
  DNM_CHECK(op->n >= 2, "real-packed form needs at least two index bits");
  std::vector<RowMask> out;
  for (const RowMask &rm : op->masks) {
    for (const RowTerm &t : rm.terms) DNM_CHECK(!t.is_imag, "operator has an imaginary matrix element: no real-packed form");
    DNM_CHECK(!rm.zero_mask_offdiag, "operator has an imaginary matrix element: no real-packed form");
    RowMask lanes;
    lanes.mask = rm.mask >> 1;
    lanes.pack_flip = (rm.mask & 1) != 0;
    if (rm.mask == 0) {
      RowMask diag;                       // what both lanes share stays the diagonal
      diag.mask = 0;
      lanes.zero_mask_offdiag = true;
      for (const RowTerm &t : rm.terms) {
        if (!(t.sign & 1)) { diag.terms.push_back({t.sign >> 1, t.coeff, 0}); continue; }
        lanes.terms.push_back({t.sign >> 1, t.coeff, 0});
        lanes.terms.push_back({t.sign >> 1, -t.coeff, 1});
      }
      if (!diag.terms.empty()) out.push_back(std::move(diag));
      if (!lanes.terms.empty()) out.push_back(std::move(lanes));
      continue;
    }
    lanes.zero_mask_offdiag = lanes.mask == 0;       // the mask flipped bit 0 only: the element's own other lane
    bool same = !lanes.pack_flip;                    // bit 0 neither flipped nor seen by a sign: both lanes get the same
    for (const RowTerm &t : rm.terms) same = same && !(t.sign & 1);      // real coefficient -- an ordinary real record
    for (const RowTerm &t : rm.terms) {
      lanes.terms.push_back({t.sign >> 1, t.coeff, 0});
      if (!same) lanes.terms.push_back({t.sign >> 1, (t.sign & 1) ? -t.coeff : t.coeff, 1});
    }
    out.push_back(std::move(lanes));
  }
  std::stable_sort(out.begin(), out.end(), [](const RowMask &a, const RowMask &b) {
    if (a.mask != b.mask) return a.mask < b.mask;
    return (int)a.zero_mask_offdiag < (int)b.zero_mask_offdiag;        // the diagonal first
  });
  op->masks.swap(out);
  op->n -= 1;
  op->packed = true;
  return 0;
}

// A mask of many terms as table records (plan.h: DevTab)?  Its terms grouped by their sign mask outside the flipped bits
// (`zs`: one record and one table per group) -- taken where that is cheaper than records of four terms (about 45 against
// 76 vector instructions each for four rows; DNM_TAB_RECORDS=0: never).
static bool table_form(const OpForm &op, const RowMask &m, std::vector<uint64_t> *zs) {
  const char *tabs_env = knob("DNM_TAB_RECORDS");
  if (tabs_env && tabs_env[0] == '0') return false;
  const int nb = __builtin_popcountll(m.mask);
  if (op.packed || m.pack_flip || nb < 1 || nb > MAXTABBITS || m.terms.size() < 5) return false;
  size_t nre = 0, nim = 0;
  zs->clear();
  for (const RowTerm &t : m.terms) {
    (t.is_imag ? nim : nre)++;
    const uint64_t z = t.sign & ~m.mask;
    if (std::find(zs->begin(), zs->end(), z) == zs->end()) zs->push_back(z);
  }
  // (masks of one record stay records: a single flip's X + iY -- the harness's long_range -- as a table of two entries made
  // that operator SLOWER, 7.11 -> 7.59 ms at L=28: a table record has its own fixed costs, the staging of the tables and the
  // kernel instance of 8 rows per thread among them)
  const size_t nq = std::max((nre + 1) / 2, (nim + 1) / 2);
  return nq >= 2 && zs->size() * 45 < nq * 76;
}

static int build_pass(const dnm_mat &A, const PassSpec &ps, PassOnDevice *out) {
  const OpForm &op = A.op;
  const Plan &pl = A.plan;
  const int B = ps.B;
  int logR = ps.logR ? ps.logR : pl.cfg.logR;
  {
    // passes with table records: rows per thread of their own (DNM_TAB_LOG_ROWS; the per-record work of a thread -- table
    // index, parity -- is shared by its rows)
    std::vector<uint64_t> zs;
    bool any = false;
    for (int idx : ps.tile_masks) any = any || table_form(op, op.masks[idx], &zs);
    for (int idx : ps.gather_masks) any = any || table_form(op, op.masks[idx], &zs);
    int want = 3;
    if (const char *e = knob("DNM_TAB_LOG_ROWS")) want = atoi(e);
    if (any && want > logR && tile_config_supported(B, want)) logR = want;
  }
  {
    // the thread part of a position has to fit a 32-bit byte offset (DevPass::pos_tmask): a tile that reaches above
    // bit 27 gives its top bits to the rows of a thread
    auto top_thread_pos = [&](int lr) {
      int c = 0, top = -1;
      for (int j = 0; j < ps.nseg; ++j)
        for (int i = 0; i < ps.seg_len[j]; ++i, ++c)
          if (c < B - lr) top = std::max(top, ps.seg_pos[j] + i);
      return top;
    };
    while (top_thread_pos(logR) >= 28 && tile_config_supported(B, logR + 1)) ++logR;
  }
  const int lognt = B - logR, R = 1 << logR;
  const int n_eff = ps.n_eff ? ps.n_eff : pl.n_loc;     // index bits this pass sweeps
  const uint64_t tb = ps.tile_bits();
  DevPass &d = out->desc;
  memset(&d, 0, sizeof(d));
  d.nseg = ps.nseg;
  int off = 0;
  for (int j = 0; j < ps.nseg; ++j) {
    d.seg_off[j] = off;
    d.seg_len[j] = ps.seg_len[j];
    d.seg_pos[j] = ps.seg_pos[j];
    off += ps.seg_len[j];
  }
  DNM_CHECK(off == B, "internal: tile segments do not add up to B");
  // block-id bits -> local index bits outside the tile.  Order (low to high):
  // three selector bits (workgroup b runs on XCD b % 8), the XCD-group bits, the rest.
  {
    std::vector<int> order;
    uint64_t gb = ps.glen ? ((((uint64_t)1 << ps.glen) - 1) << ps.gpos) : 0;
    std::vector<int> rest;
    for (int pos = 0; pos < n_eff; ++pos)
      if (!((tb >> pos) & 1) && !((gb >> pos) & 1)) rest.push_back(pos);
    size_t nsel = ps.glen ? std::min<size_t>(3, rest.size()) : 0;
    for (size_t i = 0; i < nsel; ++i) order.push_back(rest[i]);
    for (int pos = ps.gpos; pos < ps.gpos + ps.glen; ++pos) order.push_back(pos);
    for (size_t i = nsel; i < rest.size(); ++i) order.push_back(rest[i]);
    if (const char *e = knob("DNM_ORDER_WINDOW")) {     // experiments: explicit block-id bit order (low to high)
      if (ps.nseg > 1 && ps.partner < 0) {
        std::vector<int> o;
        for (const char *q = e; *q;) {
          o.push_back(atoi(q));
          while (*q && *q != ',') ++q;
          if (*q == ',') ++q;
        }
        std::vector<int> a = o, b2 = order;
        std::sort(a.begin(), a.end());
        std::sort(b2.begin(), b2.end());
        DNM_CHECK(a == b2, "DNM_ORDER_WINDOW is not a permutation of the block bits");
        order = o;
      }
    }
    DNM_CHECK((int)order.size() == n_eff - B, "internal: block bits do not add up");
    int nb = 0;
    for (size_t i = 0; i < order.size();) {
      size_t j = i + 1;
      while (j < order.size() && order[j] == order[j - 1] + 1) ++j;
      DNM_CHECK(nb < MAXBSEG, "internal: too many block segments");
      d.bseg_off[nb] = (int32_t)i;
      d.bseg_len[nb] = (int32_t)(j - i);
      d.bseg_pos[nb] = order[i];
      ++nb;
      i = j;
    }
    d.nbseg = nb;
  }
  d.sign_base = ((uint64_t)pl.rank << pl.n_loc) | ps.sign_extra;
  d.n_eff = n_eff;
  d.tile_bits = B;
  d.log_rows = logR;
  DNM_CHECK(tile_config_supported(B, logR), "unsupported tile configuration B=%d logR=%d", B, logR);
  d.accumulate = ps.accumulate ? 1 : 0;
  d.has_diag = 0;
  d.cache_policy = pl.cfg.cache_policy | (op.packed ? 256 : 0);      // bit 8: real-packed records (kernel instance)
  {
    const int S = pl.cfg.swz;
    DNM_CHECK(S == 0 || (S >= 5 && S <= 24), "swizzle shift %d out of range", S);
    d.swz_shift = S;
    auto sw = [S](uint64_t v) -> uint32_t {
      return S ? (uint32_t)(((v >> S) & (((uint64_t)1 << (S - 4)) - 1)) << 4) : 0u;
    };
    d.swz_xor_y = sw((uint64_t)ps.y_off);
    d.swz_xor_src = sw((uint64_t)ps.src_off);
    // position bits the thread part of the tile coordinate reaches (the kernel keeps them in a 32-bit byte offset)
    uint64_t tm = 0;
    int c = 0;
    for (int j = 0; j < ps.nseg; ++j)
      for (int i = 0; i < ps.seg_len[j]; ++i, ++c)
        if (c < lognt) tm |= ((uint64_t)1 << (ps.seg_pos[j] + i)) | sw((uint64_t)1 << (ps.seg_pos[j] + i));
    DNM_CHECK((tm >> 28) == 0, "internal: thread bits of the tile above bit 27 (tile %llx, %d rows per thread)",
              (unsigned long long)tb, R);
    d.pos_tmask = (uint32_t)tm;
  }

  std::vector<DevQuad> quads;
  auto empty_quad = [&]() {
    DevQuad q;
    memset(&q, 0, sizeof(q));
    return q;
  };
  auto set_slot = [&](DevQuad &q, int slot, const RowTerm &t) {
    q.sign_ext[slot] = t.sign & ~tb;
    q.sign_tile[slot] = compress_to_tile(t.sign & tb, ps);
    q.coeff[slot] = t.coeff;
  };
  // pack a list of (real) diagonal terms four to a record
  auto push_diag_list = [&](const std::vector<RowTerm> &lst) {
    for (size_t i = 0; i < lst.size(); i += 4) {
      DevQuad q = empty_quad();
      for (size_t j = i; j < lst.size() && j < i + 4; ++j) {
        set_slot(q, (int)(j - i), lst[j]);
        q.nslots = (uint32_t)(j - i + 1);
      }
      quads.push_back(q);
    }
  };

  if (ps.has_diag) {
    const RowMask *dm = nullptr;
    for (const RowMask &m : op.masks) if (m.mask == 0 && !m.zero_mask_offdiag) dm = &m;
    if (dm) {
      d.has_diag = 1;
      std::vector<RowTerm> lst;
      for (const RowTerm &t : dm->terms)
        if (compress_to_tile(t.sign & tb, ps) == 0) lst.push_back(t);
      d.dext_begin = (uint32_t)quads.size();
      push_diag_list(lst);
      d.dext_end = (uint32_t)quads.size();
      // terms inside the tile only: tabulated per tile coordinate (DNM_DIAG_TABLE=0: bucket lists as before)
      const char *dte = knob("DNM_DIAG_TABLE");
      const bool use_table = !(dte && dte[0] == '0') && B <= 13;
      if (use_table) out->h_dtile.assign((size_t)1 << B, 0.0);
      // terms that see the tile AND bits outside it, grouped by their sign mask inside the tile (DevPass::gbucket): groups of
      // three terms or more are summed over the outside bits once per workgroup (DNM_DIAG_GROUPS=0: every term per thread)
      std::vector<std::pair<uint32_t, std::vector<RowTerm>>> groups;
      {
        const char *dge = knob("DNM_DIAG_GROUPS");
        const bool grouping = !(dge && dge[0] == '0') && !op.packed && !(A.flags & DNM_MAT_USE_GLDS);
        std::vector<std::pair<uint32_t, std::vector<RowTerm>>> all;
        if (grouping)
          for (const RowTerm &t : dm->terms) {
            const uint32_t st = compress_to_tile(t.sign & tb, ps);
            if (st == 0 || (t.sign & ~tb) == 0) continue;
            auto it = std::find_if(all.begin(), all.end(), [&](const auto &g) { return g.first == st; });
            if (it == all.end()) { all.push_back({st, {}}); it = all.end() - 1; }
            it->second.push_back(t);
          }
        std::stable_sort(all.begin(), all.end(), [](const auto &a, const auto &b) { return a.second.size() > b.second.size(); });
        for (auto &g : all)
          if (g.second.size() >= 3 && groups.size() < MAXDGROUPS) groups.push_back(std::move(g));
      }
      auto grouped = [&](uint32_t st) {
        return std::any_of(groups.begin(), groups.end(), [&](const auto &g) { return g.first == st; });
      };
      for (int j = 0; j < R; ++j) {
        lst.clear();
        for (const RowTerm &t : dm->terms) {
          uint32_t st = compress_to_tile(t.sign & tb, ps);
          if (st == 0 || (int)(st >> lognt) != j) continue;
          if (use_table && (t.sign & ~tb) == 0) {
            for (uint32_t tc = 0; tc < (1u << B); ++tc)
              out->h_dtile[tc] += (__builtin_popcount(tc & st) & 1) ? -t.coeff : t.coeff;
          } else if ((t.sign & ~tb) != 0 && grouped(st)) {
            continue;
          } else {
            lst.push_back(t);
          }
        }
        d.dbucket[j] = (uint32_t)quads.size();
        push_diag_list(lst);
      }
      for (int j = R; j <= MAXR; ++j) d.dbucket[j] = (uint32_t)quads.size();
      // the groups: their term records (the outside part of every sign mask), then one record per group, by k bucket
      std::vector<std::pair<uint32_t, uint32_t>> where(groups.size());
      for (size_t g = 0; g < groups.size(); ++g) {
        std::vector<RowTerm> outside = groups[g].second;
        for (RowTerm &t : outside) t.sign &= ~tb;
        where[g].first = (uint32_t)quads.size();
        push_diag_list(outside);
        where[g].second = (uint32_t)quads.size() - where[g].first;
      }
      for (int j = 0; j < R; ++j) {
        d.gbucket[j] = (uint32_t)quads.size();
        for (size_t g = 0; g < groups.size(); ++g) {
          if ((int)(groups[g].first >> lognt) != j) continue;
          DevQuad q = empty_quad();
          q.sign_tile[0] = groups[g].first;
          q.mask_loc = where[g].first;
          q.src = where[g].second;
          q.nslots = 1;
          quads.push_back(q);
        }
      }
      for (int j = R; j <= MAXR; ++j) d.gbucket[j] = (uint32_t)quads.size();
    }
  }

  // off-diagonal masks: records of <= 2 real + <= 2 imaginary terms, sorted into
  // the kernel's loops (tile/gather x k-variant x real/complex)
  struct Rec { int loop; DevQuad q; };
  std::vector<Rec> recs;
  // masks of many terms as table records (table_form above)
  std::vector<DevTab> tabs_tile, tabs_gather;
  std::vector<double> tabvals;
  auto push_tabs = [&](const RowMask &m, uint64_t mloc, bool gather, int src) -> bool {
    std::vector<uint64_t> zs;
    if (!table_form(op, m, &zs)) return false;
    const int nb = __builtin_popcountll(m.mask);
    int pb[MAXTABBITS];
    for (int q = 0, pos = 0; pos < 64; ++pos)
      if ((m.mask >> pos) & 1ull) pb[q++] = pos;
    for (uint64_t z : zs) {
      DevTab T;
      memset(&T, 0, sizeof(T));
      T.mask_tile = compress_to_tile(mloc & tb, ps);
      T.mask_loc = (uint32_t)mloc;
      T.src = (uint32_t)src;
      T.nbits = (uint32_t)nb;
      const uint32_t zt = compress_to_tile(z & tb, ps);
      T.z_tile = zt & ((1u << lognt) - 1u);
      T.z_ext = z & ~tb;
      T.first = (uint32_t)(tabvals.size() / 2);
      for (int k = 0; k < R; ++k)
        if (__builtin_popcount((uint32_t)k & (zt >> lognt)) & 1) T.ksign |= 1u << k;
      for (int q = 0; q < nb; ++q) {
        if ((tb >> pb[q]) & 1ull) {
          const int tpos = __builtin_ctz(compress_to_tile((uint64_t)1 << pb[q], ps));
          if (tpos < lognt) {
            T.tpos |= (uint32_t)tpos << (8 * q);
            T.twid |= 1u << (8 * q);
          } else {
            T.flags |= 1u;
            for (int k = 0; k < R; ++k)
              if ((k >> (tpos - lognt)) & 1) T.ik |= (uint64_t)1 << (4 * k + q);
          }
        } else {
          T.epos |= (uint32_t)pb[q] << (8 * q);
          T.ewid |= 1u << (8 * q);
        }
      }
      for (int j = 0; j < (1 << nb); ++j) {
        uint64_t rowbits = 0;
        for (int q = 0; q < nb; ++q)
          if ((j >> q) & 1) rowbits |= (uint64_t)1 << pb[q];
        double re = 0.0, im = 0.0;
        for (const RowTerm &t : m.terms) {
          if ((t.sign & ~m.mask) != z) continue;
          const double c = (__builtin_popcountll(rowbits & t.sign & m.mask) & 1) ? -t.coeff : t.coeff;
          (t.is_imag ? im : re) += c;
        }
        tabvals.push_back(re);
        tabvals.push_back(im);
      }
      if (z == zs.back()) T.flags |= 2u;        // (the groups of a mask: consecutive records, one fetch of the partners)
      (gather ? tabs_gather : tabs_tile).push_back(T);
    }
    return true;
  };
  auto push_mask = [&](int idx, bool gather, int src) {
    const RowMask &m = op.masks[idx];
    const uint64_t mloc = m.mask & (((uint64_t)1 << n_eff) - 1);
    if (!gather) DNM_CHECK((mloc & ~tb) == 0, "internal: tile mask leaves the tile");
    if (push_tabs(m, mloc, gather, src)) return 0;
    std::vector<const RowTerm *> re, im;
    for (const RowTerm &t : m.terms) (t.is_imag ? im : re).push_back(&t);
    size_t ir = 0, ii = 0;
    while (ir < re.size() || ii < im.size()) {
      DevQuad q = empty_quad();
      q.mask_tile = compress_to_tile(mloc & tb, ps);
      q.mask_loc = (uint32_t)mloc;
      q.src = (uint32_t)src;
      q.nslots = m.pack_flip ? 1u : 0u;       // real-packed operators: a lane reads the partner's other lane
      bool kvar = false, cplx = false;
      for (int s = 0; s < 2 && ir < re.size(); ++s, ++ir) {
        set_slot(q, s, *re[ir]);
        kvar |= (q.sign_tile[s] >> lognt) != 0;
      }
      for (int s = 2; s < 4 && ii < im.size(); ++s, ++ii) {
        set_slot(q, s, *im[ii]);
        kvar |= (q.sign_tile[s] >> lognt) != 0;
        cplx = true;
      }
      int loop;
      if (gather) loop = kvar ? (cplx ? LP_GATHER_KVAR_CPLX : LP_GATHER_KVAR_REAL) : (cplx ? LP_GATHER_CPLX : LP_GATHER_REAL);
      else if (kvar) loop = cplx ? LP_TILE_KVAR_CPLX : LP_TILE_KVAR_REAL;
      else if (cplx) loop = LP_TILE_CPLX;
      else loop = (q.mask_tile >> lognt) == 0 ? LP_TILE_REAL_K0 : LP_TILE_REAL;
      recs.push_back({loop, q});
    }
    return 0;
  };
  for (int idx : ps.tile_masks) DNM_TRY(push_mask(idx, false, 0));
  for (size_t i = 0; i < ps.gather_masks.size(); ++i)
    DNM_TRY(push_mask(ps.gather_masks[i], true, ps.gather_src[i]));
  for (int lp = 0; lp < LP_COUNT; ++lp) {
    d.loop[lp] = (uint32_t)quads.size();
    for (const Rec &r : recs) if (r.loop == lp) quads.push_back(r.q);
  }
  d.loop[LP_COUNT] = (uint32_t)quads.size();

  d.nquads = (int32_t)quads.size();
  d.need_tile = (d.has_diag || !ps.tile_masks.empty()) ? 1 : 0;
  d.tab_loop[0] = 0;
  d.tab_loop[1] = (uint32_t)tabs_tile.size();
  d.tab_loop[2] = (uint32_t)(tabs_tile.size() + tabs_gather.size());
  out->h_tabs = tabs_tile;
  out->h_tabs.insert(out->h_tabs.end(), tabs_gather.begin(), tabs_gather.end());
  out->h_tabvals = tabvals;
  d.tabs = nullptr;
  d.tabvals = nullptr;
  if (!out->h_tabs.empty() && !A.host_only) {
    DNM_TRY(out->tabs.upload(out->h_tabs.data(), out->h_tabs.size() * sizeof(DevTab)));
    DNM_TRY(out->tabvals.upload(out->h_tabvals.data(), out->h_tabvals.size() * sizeof(double)));
    d.tabs = (const DevTab *)out->tabs.p;
    d.tabvals = (const double *)out->tabvals.p;
  }
  out->h_quads = quads;
  if (!A.host_only) DNM_TRY(out->quads.upload(quads.data(), quads.size() * sizeof(DevQuad)));
  d.quads = (const DevQuad *)out->quads.p;
  d.dtile = nullptr;
  if (!out->h_dtile.empty() && !A.host_only) {
    DNM_TRY(out->dtile.upload(out->h_dtile.data(), out->h_dtile.size() * sizeof(double)));
    d.dtile = (const double *)out->dtile.p;
  }
  out->partner = ps.partner;
  out->n_eff = n_eff;
  out->y_off = ps.y_off;
  out->src_off = ps.src_off;
  return 0;
}

static hipStream_t S(void *stream) { return (hipStream_t)stream; }

// d: the pass descriptor with this call's fields filled in; p: the pass it was copied from
static int launch_pass(const dnm_mat *A, const PassOnDevice &p, const DevPass &d, const void *x, void *y,
                       const void *xr, hipStream_t st, unsigned nparts = 1) {
  return launch_tile_pass(d, d.tile_bits, d.log_rows, (A->flags & DNM_MAT_USE_GLDS) != 0, p.n_eff, x, y, xr, st, nparts);
}

// partial sums a pass writes to dot_out: one per tile
static size_t pass_dot_partials(const dnm_mat *, const PassOnDevice &p) {
  return (size_t)1 << (p.n_eff - p.desc.tile_bits);
}

}  // namespace dnm

using namespace dnm;

namespace dnm {
std::vector<ScMask> sc_masks(const std::vector<int64_t> &masks, const std::vector<int64_t> &mask_offsets,
                             const std::vector<int64_t> &signs, const std::vector<double> &rcoef, int L, bool xparity) {
  std::vector<ScMask> scm(masks.size());
  for (size_t mi = 0; mi < masks.size(); ++mi) {
    ScMask &e = scm[mi];
    memset(&e, 0, sizeof(e));
    const uint64_t mask = (uint64_t)masks[mi];
    e.dead = __builtin_popcountll(mask) & 1;
    if (xparity && L >= 3 && __builtin_popcountll(mask) == L - 2 && !((mask >> (L - 1)) & 1ull)) {
      // a hop between spin i and spin L-1 times the global flip (XParity.reduce_msc): every spin but those two
      const uint64_t miss = ~mask & ((((uint64_t)1 << (L - 1)) - 1));
      const int i = __builtin_ctzll(miss);
      const uint64_t pairbits = ((uint64_t)1 << i) | ((uint64_t)1 << (L - 1));
      bool local = true;
      for (int64_t t = mask_offsets[mi]; t < mask_offsets[mi + 1]; ++t)
        if ((uint64_t)signs[t] & ~pairbits) local = false;
      if (!local) continue;
      e.pair = 2;
      e.lo = i;
      e.hi = L - 1;
      for (int64_t t = mask_offsets[mi]; t < mask_offsets[mi + 1]; ++t) {
        // the column state keeps spin i down and spin L-1 up; the sign masks do not meet the mask: a real element
        const double c = (((uint64_t)signs[t] >> i) & 1) ? -rcoef[t] : rcoef[t];
        e.up_re += c;
        e.dn_re += c;
      }
      continue;
    }
    if (__builtin_popcountll(mask) != 2) continue;
    const int lo = __builtin_ctzll(mask), hi = 63 - __builtin_clzll(mask);
    bool local = true;
    for (int64_t t = mask_offsets[mi]; t < mask_offsets[mi + 1]; ++t)
      if ((uint64_t)signs[t] & ~mask) local = false;
    if (!local) continue;
    e.pair = 1;
    e.fast = hi == lo + 1;
    e.lo = lo;
    e.hi = hi;
    for (int64_t t = mask_offsets[mi]; t < mask_offsets[mi + 1]; ++t) {
      const uint64_t sg = (uint64_t)signs[t];
      const double rc = rcoef[t];
      const bool imag = parity64(mask & sg);
      // column state (bra) carries the moved spin: bit hi for an up hop, bit lo for a down hop
      const double up = ((sg >> hi) & 1) ? -rc : rc;
      const double dn = ((sg >> lo) & 1) ? -rc : rc;
      (imag ? e.up_im : e.up_re) += up;
      (imag ? e.dn_im : e.dn_re) += dn;
    }
  }
  return scm;
}
}  // namespace dnm

extern "C" {

const char *dnm_last_error(void) { return g_err.c_str(); }
int dnm_version(void) { return 100; }

int dnm_device_count(int *count) {
  DNM_CHECK(count, "null pointer");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); }
  return 0;
}
int dnm_set_device(int device) { DNM_HIP(hipSetDevice(device)); return 0; }
int dnm_malloc(void **dptr, size_t bytes) { DNM_HIP(hipMalloc(dptr, bytes ? bytes : 16)); return 0; }
int dnm_free(void *dptr) { DNM_HIP(hipFree(dptr)); return 0; }
int dnm_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream) {
  DNM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream)));
  DNM_HIP(hipStreamSynchronize(S(stream)));
  return 0;
}
int dnm_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream) {
  DNM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream)));
  DNM_HIP(hipStreamSynchronize(S(stream)));
  return 0;
}
int dnm_stream_synchronize(void *stream) { DNM_HIP(hipStreamSynchronize(S(stream))); return 0; }

// ---- subspaces --------------------------------------------------------------
int dnm_subspace_dim(const dnm_subspace *s, int64_t *dim) {
  SubView v{};
  DNM_TRY(view_from_c(s, &v));
  *dim = sub_dim(v);
  return 0;
}

int dnm_idx_to_state(const dnm_subspace *s, int64_t n, const int64_t *idxs, int64_t *states) {
  SubView v{};
  DNM_TRY(view_from_c(s, &v));
  const int64_t dim = sub_dim(v);
  for (int64_t i = 0; i < n; ++i) {
    // the reference raises for out-of-range indices (bsubspace.pyx:170-173)
    DNM_CHECK(idxs[i] >= 0 && idxs[i] < dim, "index %lld out of bounds for subspace of dimension %lld",
              (long long)idxs[i], (long long)dim);
    states[i] = sub_i2s(idxs[i], v);
  }
  return 0;
}

int dnm_state_to_idx(const dnm_subspace *s, int64_t n, const int64_t *states, int64_t *idxs) {
  SubView v{};
  DNM_TRY(view_from_c(s, &v));
  for (int64_t i = 0; i < n; ++i) idxs[i] = sub_s2i(states[i], v);
  return 0;
}

// ---- CheckConserves ---------------------------------------------------------------
int dnm_check_conserves(int64_t nmasks, const int64_t *masks, const int64_t *mask_offsets,
                        const int64_t *signs, const double *coeffs, const dnm_subspace *left,
                        const dnm_subspace *right, int xparity, int *result, void *stream) {
  DNM_CHECK(result && (nmasks == 0 || (masks && mask_offsets && signs && coeffs)), "null argument");
  SubOwned l, r;
  DNM_TRY(l.init(left, true));
  DNM_TRY(r.init(right, true));
  const int64_t nterms = nmasks ? mask_offsets[nmasks] : 0;
  std::vector<double> re(nterms), im(nterms);
  for (int64_t t = 0; t < nterms; ++t) { re[t] = coeffs[2 * t]; im[t] = coeffs[2 * t + 1]; }
  DevBuf dm, doff, ds, dre, dim_, dbad;
  DNM_TRY(dm.upload(masks, (size_t)nmasks * 8));
  DNM_TRY(doff.upload(mask_offsets, (size_t)(nmasks + 1) * 8));
  DNM_TRY(ds.upload(signs, (size_t)nterms * 8));
  DNM_TRY(dre.upload(re.data(), (size_t)nterms * 8));
  DNM_TRY(dim_.upload(im.data(), (size_t)nterms * 8));
  int zero = 0;
  DNM_TRY(dbad.upload(&zero, sizeof(int)));
  DevMsc msc{(int32_t)nmasks, (const int64_t *)dm.p, (const int64_t *)doff.p, (const int64_t *)ds.p,
             (const double *)dre.p};
  // XParity: the columns are the first half of the parent's (bpetsc_template_2.c:1005-1008)
  const int64_t ncols = xparity ? r.host.dim / 2 : r.host.dim;
  DNM_TRY(launch_conserves(msc, (const double *)dim_.p, l.dev, r.dev, ncols, (int *)dbad.p, S(stream)));
  int bad = 0;
  DNM_TRY(dnm_memcpy_d2h(&bad, dbad.p, sizeof(int), stream));
  *result = bad ? 0 : 1;
  return 0;
}

// ---- reduced density matrix -----------------------------------------------------------
static DevBuf g_rdm_scratch;     // released by dnm_release_workspace()
}  // extern "C"
namespace dnm {
int rdm_release_scratch() {
  g_rdm_scratch.release();
  return 0;
}
}  // namespace dnm
extern "C" {

int dnm_reduced_density_matrix(const void *x, const dnm_subspace *sub, int keep_size, const int64_t *keep,
                               void *rho, void *stream) {
  DNM_CHECK(x && sub && rho && keep_size >= 0 && (keep_size == 0 || keep), "null argument");
  SubOwned s;
  DNM_TRY(s.init(sub, true));
  const int L = s.host.L;
  DNM_CHECK(keep_size <= L, "more kept spins than spins");
  DNM_CHECK(keep_size <= 15, "reduced density matrix of %d spins (4^%d entries) is too large", keep_size, keep_size);
  for (int i = 0; i < keep_size; ++i) {
    DNM_CHECK(keep[i] >= 0 && keep[i] < L, "kept spin index %lld out of range [0, %d)", (long long)keep[i], L);
    // bpetsc_template_1.c:117-121
    DNM_CHECK(i == 0 || keep[i] > keep[i - 1], "keep array must be strictly increasing");
  }
  RdmGeom geo;
  memset(&geo, 0, sizeof(geo));
  geo.k = keep_size;
  geo.L = L;
  uint64_t keepmask = 0;
  for (int i = 0; i < keep_size; ++i) keepmask |= (uint64_t)1 << keep[i];
  for (int pos = 0; pos < L;) {     // runs of kept / traced positions
    const bool kept = (keepmask >> pos) & 1;
    int end = pos;
    while (end < L && (((keepmask >> end) & 1) != 0) == kept) ++end;
    if (kept) {
      geo.klen[geo.nseg_keep] = (int8_t)(end - pos);
      geo.kpos[geo.nseg_keep++] = (int8_t)pos;
    } else {
      geo.tlen[geo.nseg_tr] = (int8_t)(end - pos);
      geo.tpos[geo.nseg_tr++] = (int8_t)pos;
    }
    pos = end;
  }
  int logtm, ntiles, nsplit;
  int64_t cps;
  size_t pbytes;
  rdm_plan(geo, &logtm, &ntiles, &nsplit, &cps, &pbytes);
  // scratch for the partial tiles: small ones are kept between calls (hipMalloc costs more than the kernel)
  DevBuf &cached = g_rdm_scratch;
  DevBuf big;
  void *scratch = nullptr;
  if (pbytes <= ((size_t)1 << 30)) {
    if (cached.bytes < pbytes) {
      cached.release();
      DNM_TRY(cached.alloc(pbytes));
    }
    scratch = cached.p;
  } else {
    DNM_TRY(big.alloc(pbytes));
    scratch = big.p;
  }
  DNM_TRY(launch_rdm(x, s.dev, geo, scratch, rho, S(stream)));
  DNM_HIP(hipStreamSynchronize(S(stream)));     // `big` is released on return
  return 0;
}

// ---- shell matrix -------------------------------------------------------------
// Decide whether the SpinConserve block kernel runs and build its table.  DNM_SC_BLOCK = 0 (row kernel),
// 10 / 13 (low bits per block); default: 13 when the blocks are large enough to fill a workgroup.
static int setup_sc_block(dnm_mat *A) {
  A->scblock.lb = 0;
  if (!A->sc_pair || A->m_local <= 0) return 0;
  const SubView &h = A->left.host;
  const int L = h.L, k = h.k;
  int lb = -1;
  if (const char *e = knob("DNM_SC_BLOCK")) lb = atoi(e);
  if (lb == 0) return 0;
  const bool forced = lb > 0;
  if (!forced) lb = 13;
  if ((int64_t)A->masks.size() > sc_block_max_masks()) return 0;     // one lane per mask in the block kernel
  DNM_CHECK(sc_block_supported(lb), "DNM_SC_BLOCK=%d: no such kernel instance (0, 10, 13)", lb);
  if (L <= lb || L - lb > 48) return 0;
  if (!forced && (A->M >> (L - lb)) < 256) return 0;     // blocks too small on average: one row per thread instead
  if (!forced && 2 * A->sc_nfast < (int)A->masks.size()) return 0;   // mostly non-chain masks: they take the per-row path anyway
  std::vector<uint16_t> tab((size_t)1 << lb);
  int cnt[18] = {0};
  for (int v = 0; v < (1 << lb); ++v) ++cnt[__builtin_popcount(v) + 1];
  for (int j = 0; j < 17; ++j) { cnt[j + 1] += cnt[j]; A->scblock.off[j] = cnt[j]; }
  A->scblock.off[17] = 1 << lb;
  int fill[17];
  for (int j = 0; j < 17; ++j) fill[j] = A->scblock.off[j];
  for (int v = 0; v < (1 << lb); ++v) tab[fill[__builtin_popcount(v)]++] = (uint16_t)v;
  DNM_TRY(A->d_scblock.upload(tab.data(), tab.size() * sizeof(uint16_t)));
  A->scblock.lowtab = (const uint16_t *)A->d_scblock.p;
  int64_t hf = sub_i2s(A->row0, h) >> lb, hl = sub_i2s(A->row0 + A->m_local - 1, h) >> lb;
  A->scblock.swizzle = 0;
  if (hl - (hf & ~(int64_t)511) + 1 >= 1024) {
    A->scblock.swizzle = 1;
    hf &= ~(int64_t)511;
  }
  A->scblock.h_first = hf;
  A->scblock.h_last = hl;
  A->scblock.lb = lb;
  A->scblock.perm = nullptr;
  A->scblock.nperm = 0;
  // Block order (DNM_SC_ORDER=g, 0 = ascending H): blocks of equal size run together -- ordered by the popcount
  // of H, then by the bits of H above the lowest g -- so that the workgroups resident at one time do equal work
  // and stay in step, and the blocks an XCD holds are siblings under the g-1 lowest high bonds; groups go
  // round-robin to the XCDs.  Measured on MI355X, L=32 k=16: 17.6 ms ascending, 15.3 (g=1), 14.4 (g=6), 15.9 (g=8).
  int g = (hl - hf + 1 >= 1024) ? 6 : 0;
  if (const char *e = knob("DNM_SC_ORDER")) g = atoi(e);
  // DNM_SC_CHUNK=c (experiment): the order above inside chunks of 2^c consecutive high parts, chunk after chunk --
  // partners under the c-1 lowest high bonds then lie in the same chunk, i.e. within what the Infinity Cache holds
  int chunk = 0;
  if (const char *e = knob("DNM_SC_CHUNK")) chunk = atoi(e);
  if (chunk <= 0 || chunk > 40) chunk = 62;
  if (g > 0 && hl - hf + 1 < ((int64_t)1 << 31)) {
    const int64_t span = hl - hf + 1;
    std::vector<uint32_t> ord;
    ord.reserve((size_t)span);
    for (int64_t e = 0; e < span; ++e) {
      const int kl = k - __builtin_popcountll((uint64_t)(hf + e));
      if (kl >= 0 && kl <= lb) ord.push_back((uint32_t)e);
    }
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
      const uint64_t ha = (uint64_t)(hf + a), hb = (uint64_t)(hf + b);
      const int pa = __builtin_popcountll(ha), pb = __builtin_popcountll(hb);
      if ((ha >> chunk) != (hb >> chunk)) return (ha >> chunk) < (hb >> chunk);
      if (pa != pb) return pa < pb;
      if ((ha >> g) != (hb >> g)) return (ha >> g) < (hb >> g);
      return ha < hb;
    });
    std::vector<std::vector<uint32_t>> lists(8);
    size_t i = 0;
    int64_t q = 0;
    while (i < ord.size()) {
      size_t j = i;
      const uint64_t h0 = (uint64_t)(hf + ord[i]);
      while (j < ord.size() && ((uint64_t)(hf + ord[j]) >> g) == (h0 >> g) &&
             __builtin_popcountll((uint64_t)(hf + ord[j])) == __builtin_popcountll(h0)) ++j;
      auto &dst = lists[(size_t)(q++ & 7)];
      dst.insert(dst.end(), ord.begin() + i, ord.begin() + j);
      i = j;
    }
    size_t longest = 0;
    for (auto &l : lists) longest = std::max(longest, l.size());
    std::vector<uint32_t> perm(longest * 8, 0xffffffffu);
    for (size_t x = 0; x < 8; ++x)
      for (size_t t = 0; t < lists[x].size(); ++t) perm[t * 8 + x] = lists[x][t];
    DNM_TRY(A->d_scperm.upload(perm.data(), perm.size() * sizeof(uint32_t)));
    A->scblock.perm = (const uint32_t *)A->d_scperm.p;
    A->scblock.nperm = (int64_t)perm.size();
  }
  return 0;
}

int dnm_mat_create(int64_t nmasks, const int64_t *masks, const int64_t *mask_offsets,
                   const int64_t *signs, const double *coeffs, const dnm_subspace *left,
                   const dnm_subspace *right, int xparity, int flags, const dnm_partition *part,
                   dnm_mat **out) {
  DNM_CHECK(out, "null output handle");
  *out = nullptr;
  DNM_CHECK(nmasks >= 0 && (nmasks == 0 || (masks && mask_offsets && signs && coeffs)),
            "null operator arrays");
  std::unique_ptr<dnm_mat> A(new dnm_mat());
  A->xparity = xparity != 0;
  A->flags = flags;
  A->host_only = (flags & DNM_MAT_HOST_ONLY) != 0;
  const int64_t nterms = nmasks ? mask_offsets[nmasks] : 0;
  A->masks.assign(masks, masks + nmasks);
  if (nmasks) A->mask_offsets.assign(mask_offsets, mask_offsets + nmasks + 1);
  else A->mask_offsets.assign(1, 0);
  A->signs.assign(signs, signs + nterms);
  A->real_coeffs.resize(nterms);
  for (int64_t i = 1; i < nmasks; ++i)
    DNM_CHECK(masks[i] > masks[i - 1], "masks must be sorted and unique");
  for (int64_t t = 0; t < nterms; ++t) {
    // one double per term: the real part if non-zero, else the imaginary part
    const double re = coeffs[2 * t], im = coeffs[2 * t + 1];
    A->real_coeffs[t] = (re != 0.0) ? re : im;
  }
  DNM_TRY(A->left.init(left, !A->host_only));
  DNM_TRY(A->right.init(right, !A->host_only));
  DNM_CHECK(A->left.host.L == A->right.host.L, "left and right subspaces have different L");
  A->M = A->left.host.dim;
  A->N = A->right.host.dim;
  if (A->xparity) {
    // The operator has been rewritten by XParity.reduce_msc (subspaces.py:632-674) and the basis is
    // the first half of the parent's: halving the dimensions is all the backend does
    // (bpetsc_template_2.c:78-85,223-230).  No mask may flip spin L-1 any more.
    DNM_CHECK(A->M % 2 == 0 && A->N % 2 == 0, "XParity needs parent subspaces of even dimension");
    const int64_t top = (int64_t)1 << (A->left.host.L - 1);
    for (int64_t i = 0; i < nmasks; ++i)
      DNM_CHECK(!(masks[i] & top), "XParity: mask %lld flips spin L-1 (operator not reduced by XParity.reduce_msc)",
                (long long)masks[i]);
    A->M /= 2;
    A->N /= 2;
  }
  A->rank = part ? part->rank : 0;
  A->nranks = part ? part->nranks : 1;
  DNM_CHECK(A->nranks >= 1 && A->rank >= 0 && A->rank < A->nranks, "bad partition (rank %d of %d)", A->rank,
            A->nranks);

  const int lt = A->left.host.type, rt = A->right.host.type;
  A->hypercube = (lt == rt) && (lt == DNM_FULL || lt == DNM_PARITY);
  A->sc_pair = lt == DNM_SPIN_CONSERVE && rt == DNM_SPIN_CONSERVE && A->left.host.k == A->right.host.k &&
               !(flags & DNM_MAT_FORCE_GATHER);
  if (A->nranks > 1 && A->hypercube && ((A->nranks & (A->nranks - 1)) != 0 || A->M % A->nranks != 0)) {
    // Full / Parity on a rank count that is not a power of two: blocks are not subcubes, so the XOR-partner
    // exchange does not apply -- rows are split as PetscSplitOwnership does and the generic row kernel reads
    // its columns through a window (MatMult_CPU_General's MPI branch, bpetsc_template_2.c:413-504)
    A->hypercube = false;
  }
  if (A->nranks > 1 && !A->hypercube) {
    // rows of a window partition are index-ordered blocks of any size: swizzled blocks need power-of-two sizes
    DNM_CHECK(A->left.host.swz == 0 || (A->M % A->nranks == 0 && ((A->M / A->nranks) & (A->M / A->nranks - 1)) == 0),
              "swizzled left vectors need power-of-two blocks (use vec_swizzle = 0 for this partition)");
  }
  // PetscSplitOwnership: M / P rows each, the first M % P ranks one more
  {
    const int64_t q = A->M / A->nranks, rem = A->M % A->nranks;
    A->m_local = q + (A->rank < rem ? 1 : 0);
    A->row0 = (int64_t)A->rank * q + std::min<int64_t>(A->rank, rem);
    const int64_t qn = A->N / A->nranks, remn = A->N % A->nranks;
    A->n_local = qn + (A->rank < remn ? 1 : 0);
    A->rows_local = A->m_local;
  }
  // SpinConserve pair whose vectors are in the internal layout: the two-pass / row kernels of sc3_kernels.hip.  Any
  // other use of such a subspace (another partner, XParity, several ranks) works in reference order: the caller
  // converts (dnm_mat_layouts tells).
  // (XParity on top: one rank, half filling -- the vectors are the blocks whose top bit is clear, the first half of the
  // layout as of the reference order)
  A->use_sc3 = A->sc_pair && A->left.host.sc3 != 0 && A->left.host.sc3 == A->right.host.sc3 &&
               A->left.host.L == A->right.host.L &&
               (!A->xparity || (A->nranks == 1 && 2 * A->left.host.k == A->left.host.L &&
                                sc3_instance(sc3_code_a(A->left.host.sc3), sc3_code_w(A->left.host.sc3))));

  std::vector<ScMask> scm;
  if (lt == DNM_SPIN_CONSERVE && rt == DNM_SPIN_CONSERVE)
    scm = sc_masks(A->masks, A->mask_offsets, A->signs, A->real_coeffs, A->left.host.L, A->xparity);
  // tables for the generic kernels (always: norm and diagonal use them)
  if (!A->host_only) {
  DNM_TRY(A->d_masks.upload(A->masks.data(), A->masks.size() * 8));
  DNM_TRY(A->d_offsets.upload(A->mask_offsets.data(), A->mask_offsets.size() * 8));
  DNM_TRY(A->d_signs.upload(A->signs.data(), A->signs.size() * 8));
  DNM_TRY(A->d_rcoeffs.upload(A->real_coeffs.data(), A->real_coeffs.size() * 8));
  A->dmsc.nmasks = (int32_t)nmasks;
  A->dmsc.masks = (const int64_t *)A->d_masks.p;
  A->dmsc.mask_offsets = (const int64_t *)A->d_offsets.p;
  A->dmsc.signs = (const int64_t *)A->d_signs.p;
  A->dmsc.real_coeffs = (const double *)A->d_rcoeffs.p;
  if (lt == DNM_SPIN_CONSERVE && rt == DNM_SPIN_CONSERVE) {
    A->sc_nfast = 0;
    for (const ScMask &e : scm) A->sc_nfast += e.fast ? 1 : 0;
    DNM_TRY(A->d_scmasks.upload(scm.data(), scm.size() * sizeof(ScMask)));
    // 16-bit patterns grouped by popcount, ascending inside a group (colex order = numeric order)
    std::vector<uint16_t> low(65536);
    int cnt[18] = {0};
    for (int v = 0; v < 65536; ++v) ++cnt[__builtin_popcount(v) + 1];
    for (int j = 0; j < 17; ++j) { cnt[j + 1] += cnt[j]; A->sclow.off[j] = cnt[j]; }
    A->sclow.off[17] = 65536;
    int fill[17];
    for (int j = 0; j < 17; ++j) fill[j] = A->sclow.off[j];
    for (int v = 0; v < 65536; ++v) low[fill[__builtin_popcount(v)]++] = (uint16_t)v;
    DNM_TRY(A->d_sclow.upload(low.data(), low.size() * sizeof(uint16_t)));
    A->sclow.tab = (const uint16_t *)A->d_sclow.p;
    if (!A->use_sc3) DNM_TRY(setup_sc_block(A.get()));
  }
  }
  if (A->use_sc3) {
    const SubView &h = A->left.host;
    const Sc3Layout *ly = sc3_get(h.L, h.k, sc3_code_a(h.sc3), sc3_code_w(h.sc3), !A->host_only, sc3_code_order(h.sc3));
    DNM_CHECK(ly, "could not build the SpinConserve vector layout");
    A->sc3.reset(new Sc3Mat());
    // partitioned: whole T blocks per rank (a contiguous range of both the internal layout and the reference order)
    std::vector<uint32_t> Tb = sc3_partition(*ly, A->nranks);
    if (A->xparity) Tb = {0u, 1u << (ly->host.t - 1)};
    const bool want_real = (flags & DNM_MAT_REAL_PACKED) != 0;
    // site relabelling of the vectors (dnm_subspace.site_perm): the kernels of the layout see the operator in it
    Sc3Perm P, Pr;
    DNM_CHECK(sc3_perm_make(left->site_perm, h.L, &P) && sc3_perm_make(right->site_perm, h.L, &Pr),
              "site_perm is not a permutation of the %d spins", h.L);
    DNM_CHECK(memcmp(P.to_int, Pr.to_int, sizeof P.to_int) == 0, "left and right vectors must share their site relabelling");
    DNM_CHECK(!P.on || A->nranks == 1, "a relabelled SpinConserve layout is not partitioned over ranks");
    DNM_CHECK(!A->xparity || P.to_int[h.L - 1] == h.L - 1, "XParity: the site relabelling must keep spin L-1 in place");
    A->sc3->perm = P;
    std::vector<int64_t> pmasks = A->masks, poffs = A->mask_offsets, psigns = A->signs;
    std::vector<double> pcoef = A->real_coeffs;
    if (P.on) {
      // masks and signs bit by bit into the layout's labelling; the mask groups sorted again (terms keep their order
      // inside a group; a coefficient is unchanged: it multiplies the same product of Pauli matrices)
      std::vector<int64_t> order((size_t)nmasks);
      std::vector<uint64_t> nm((size_t)nmasks);
      for (int64_t i = 0; i < nmasks; ++i) {
        order[(size_t)i] = i;
        nm[(size_t)i] = sc3_permute((uint64_t)A->masks[(size_t)i], P.to_int, h.L);
      }
      std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return nm[(size_t)x] < nm[(size_t)y]; });
      pmasks.clear(); psigns.clear(); pcoef.clear();
      poffs.assign(1, 0);
      for (int64_t q = 0; q < nmasks; ++q) {
        const int64_t i = order[(size_t)q];
        pmasks.push_back((int64_t)nm[(size_t)i]);
        for (int64_t t = A->mask_offsets[(size_t)i]; t < A->mask_offsets[(size_t)i + 1]; ++t) {
          psigns.push_back((int64_t)sc3_permute((uint64_t)A->signs[(size_t)t], P.to_int, h.L));
          pcoef.push_back(A->real_coeffs[(size_t)t]);
        }
        poffs.push_back((int64_t)psigns.size());
      }
      scm = sc_masks(pmasks, poffs, psigns, pcoef, h.L, A->xparity);
      if (!A->host_only) {           // the row kernel's tables in the layout's labelling
        DNM_TRY(A->d_pmasks.upload(pmasks.data(), pmasks.size() * 8));
        DNM_TRY(A->d_poffsets.upload(poffs.data(), poffs.size() * 8));
        DNM_TRY(A->d_psigns.upload(psigns.data(), psigns.size() * 8));
        DNM_TRY(A->d_prcoeffs.upload(pcoef.data(), pcoef.size() * 8));
      }
    }
    A->dmsc_sc3 = A->dmsc;
    if (P.on && !A->host_only) {
      A->dmsc_sc3.masks = (const int64_t *)A->d_pmasks.p;
      A->dmsc_sc3.mask_offsets = (const int64_t *)A->d_poffsets.p;
      A->dmsc_sc3.signs = (const int64_t *)A->d_psigns.p;
      A->dmsc_sc3.real_coeffs = (const double *)A->d_prcoeffs.p;
    }
    DNM_TRY(A->sc3->init(ly, pmasks, poffs, psigns, pcoef, scm, !A->host_only, Tb[A->rank],
                         Tb[A->rank + 1], want_real));
    if (const char *e = knob("DNM_SC3_TILED")) if (e[0] == '0') A->sc3->tiled = false;     // tests: the row kernel
    if (const char *e = knob("DNM_SC3_DIAG")) if (e[0] == 'c' && A->sc3->diag_mode == 2) A->sc3->diag_mode = 1;
    // (timing probe, WRONG results: the passes without any diagonal -- what dropping the cached diagonal's 8 B/row could save)
    if (const char *e = knob("DNM_SC3_DIAG")) if (e[0] == 'n') A->sc3->diag_mode = 0;
    int64_t is, il, ns, nl;
    sc3_range(*ly, Tb[A->rank], Tb[A->rank + 1], &is, &il, &ns, &nl);
    A->m_local = A->n_local = il;          // what the vector kernels sweep: rows + padding
    A->rows_local = nl;
    A->row0 = ns;                          // first row in reference order (norm / diagonal kernels)
    if (want_real) {
      // real vectors in the same positions of the layout, one double each: a vector is il / 2 complex128 elements for
      // everything that sweeps it (the Krylov kernels); the two tiled passes only (chain operators, real symmetric)
      // Partitions: every position the C ABI speaks of for such a handle -- ownership, column windows, chunk maps,
      // window starts -- counts complex128 ELEMENTS (pairs of entries; blocks of equal top bits start at multiples of 8
      // entries), so the caller's exchange code is the one it runs for complex vectors, on half the bytes
      DNM_CHECK(A->sc3->tiled && A->sc3->sym,
                "operator has an imaginary matrix element or is not a sum of pair hops: no real-packed form in this layout");
      A->real_packed = true;
      A->m_local = A->n_local = il / 2;
    }
  } else if (flags & DNM_MAT_REAL_PACKED) {
    DNM_CHECK(A->hypercube, "real-packed operators: Full / Parity pairs, or a SpinConserve pair in the internal layout");
  }

  if (A->hypercube) {
    DNM_TRY(build_opform(*A, &A->op));
    if (flags & DNM_MAT_REAL_PACKED) {
      // (partitions: the rank bits are the top index bits, the packed bit is bit 0 -- the partner exchange of the packed
      // operator is that of an operator on one bit less; the transposed exchange is not built for it)
      // (XParity on top: the reduced operator is packed like any other -- its flip-composed masks reach index bit 0, the
      // lane, like every odd mask -- and loses its top index bit below)
      DNM_CHECK(A->M == A->N, "real-packed operators: square");
      DNM_TRY(pack_opform(&A->op));
      A->real_packed = true;
      A->M /= 2; A->N /= 2; A->m_local /= 2; A->n_local /= 2;         // complex128 elements, two amplitudes each
    }
    if (A->xparity) {
      // rows and columns have the top index bit clear: the hypercube loses one dimension
      const uint64_t topbit = (uint64_t)1 << (A->op.n - 1);
      for (RowMask &rm : A->op.masks) {
        DNM_CHECK(!(rm.mask & topbit), "internal: XParity mask reaches the top index bit");
        for (RowTerm &t : rm.terms) t.sign &= ~topbit;
      }
      A->op.n -= 1;
    }
    PlanConfig cfg = plan_config_from_env();
    if (const int fa = (flags >> DNM_MAT_AMIN_SHIFT) & 0xff) cfg.amin = fa;      // caller-chosen run length of window tiles
    DNM_CHECK(A->left.host.swz == A->right.host.swz, "left and right vectors of a Full/Parity pair must share a layout");
    cfg.swz = A->left.host.swz;
    if (cfg.logR < 0) {
      // measured: in index order 16 rows per thread pay from 2^26 local amplitudes on (profiles/r01_sweep6.txt);
      // with swizzled vectors 8 rows beat 16 at every size (profiles/r02_exp3_v2.txt), and once the window pass
      // runs first and the accumulating pass carries the diagonal, 4 rows per thread (two 1024-thread workgroups,
      // 32 waves per CU) beat 8: L=30 16.9 -> 16.5 ms, 2-3 % from 2^26 amplitudes on
      // (profiles/r02_exp46_rows4_series.txt, r02_exp47_rows4_sizes.txt)
      int nl = A->op.n;
      for (int r = A->nranks; r > 1; r >>= 1) --nl;
      cfg.logR = cfg.swz ? 2 : (nl >= 26 ? 4 : 3);
    }
    if (cfg.diag_last < 0) cfg.diag_last = cfg.swz ? 1 : 0;
    if (cfg.window_first < 0) {
      // which pass writes y and which adds to it: the window pass is bound by its bytes (48 B/amp when it
      // accumulates, at the streaming rate), the contiguous pass by its records (8.0 ms for 32.9 B/amp) -- so the y
      // read belongs to the latter.  With swizzled vectors (in index order the window pass's gathers do not merge
      // and it is the slow one either way, profiles/r01_prof_multi18.txt): L=30 17.5 -> 17.0 ms, L=28 4.43 -> 4.22,
      // L=26 1.09 -> 1.07; level or worse while both vectors fit in the Infinity Cache
      // (profiles/r02_exp41_pass_order.txt, r02_exp42_window_first_wide.txt)
      int nl = A->op.n;
      for (int r = A->nranks; r > 1; r >>= 1) --nl;
      cfg.window_first = (cfg.swz && nl >= 25) ? 1 : 0;
    }
    if (!tile_config_supported(cfg.B, cfg.logR)) {
      set_error("unsupported tile configuration B=%d logR=%d", cfg.B, cfg.logR);
      return 1;
    }
    DNM_TRY(make_plan(A->op, A->rank, A->nranks, cfg, &A->plan));
    if (flags & DNM_MAT_FORCE_GATHER) A->plan.use_tiled = false;
    DNM_CHECK(!A->real_packed || A->plan.use_tiled, "real-packed operators run on the tiled kernel only (vector too small)");
    if (A->nranks > 1 && !A->plan.use_tiled) {
      // blocks smaller than one tile (or DNM_MAT_FORCE_GATHER): the window partition of the row kernel
      A->plan.remote.clear();
      A->plan.sends.clear();
    }
    if (A->plan.use_tiled) {
      for (const PassSpec &ps : A->plan.local) {
        std::unique_ptr<PassOnDevice> p(new PassOnDevice());
        DNM_TRY(build_pass(*A, ps, p.get()));
        A->local_passes.push_back(std::move(p));
      }
      for (const PassSpec &ps : A->plan.remote) {
        std::unique_ptr<PassOnDevice> p(new PassOnDevice());
        DNM_TRY(build_pass(*A, ps, p.get()));
        A->remote_passes.push_back(std::move(p));
      }
    }
  }
  *out = A.release();
  return 0;
}

int dnm_mat_destroy(dnm_mat *A) {
  delete A;
  return 0;
}

// Exchange scheme of a partitioned Full / Parity operator.  The transposed exchange (backend.transpose_split is the
// host-side statement of the same split, tests/test_transpose_exchange.py compares the two): a rank's block holds the
// n = L - p low index bits, the rank number is the p top ones.  A term that flips a top bit couples blocks of different
// ranks; instead of shipping a partner block per such mask (bpetsc_template_2.c:787-879 scatters the needed entries) the
// state is redistributed ONCE so that the top bits become local: layout B swaps bits [n, n + p) with the local field
// F = [f, f + p), f = n - 1 - p (right below the top local bit, which the boundary bond touches).  In layout B every
// term that flips a top bit is rank-local provided it leaves F alone -- the same MSC term with the two fields swapped.
int dnm_mat_set_exchange(dnm_mat *A, int scheme, int *chosen) {
  DNM_CHECK(A && chosen, "null argument");
  DNM_CHECK(scheme == DNM_EXCHANGE_AUTO || scheme == DNM_EXCHANGE_PARTNER || scheme == DNM_EXCHANGE_TRANSPOSE,
            "unknown exchange scheme %d", scheme);
  delete A->tr_lo; delete A->tr_hi;
  A->tr_lo = A->tr_hi = nullptr;
  A->tr_f = -1;
  *chosen = DNM_EXCHANGE_PARTNER;
  const int P = A->nranks;
  if (scheme == DNM_EXCHANGE_PARTNER || P < 2 || (scheme == DNM_EXCHANGE_AUTO && P < 4)) return 0;
  const SubView &lh = A->left.host, &rh = A->right.host;
  if (!(A->hypercube && A->plan.use_tiled && !A->xparity && lh.type == rh.type && lh.L == rh.L &&
        (lh.type == DNM_FULL || lh.space == rh.space) && lh.swz == rh.swz))
    return 0;
  int p = 0;
  while ((1 << p) < P) ++p;
  const int shift = lh.type == DNM_PARITY ? 1 : 0;      // Parity: index = configuration >> 1, index bit j is spin j + 1
  const int packed = A->real_packed ? 1 : 0;            // index bit 0 is the lane of an element: F must leave it alone
  const int n = (int)lh.L - shift - p, f = n - 1 - p;
  if ((1 << p) != P || f < packed) return 0;
  if (lh.swz && f - packed < 2 * lh.swz - 4) return 0;  // pieces of 2^f amplitudes keep their internal order in both layouts
  const int fs = f + shift, ns = n + shift;             // the two fields as spins
  const int64_t fld = P - 1;
  struct Term { int64_t m, s; double c; };
  std::vector<Term> lo, hi;
  for (size_t i = 0; i < A->masks.size(); ++i)
    for (int64_t t = A->mask_offsets[i]; t < A->mask_offsets[i + 1]; ++t) {
      const int64_t m = A->masks[i], s = A->signs[(size_t)t];
      if ((m >> ns) == 0) { lo.push_back({m, s, A->real_coeffs[(size_t)t]}); continue; }
      if ((m >> fs) & fld) return 0;                    // a term that flips a rank bit AND the field it would move to
      auto swap = [&](int64_t v) {
        const int64_t d = ((v >> fs) ^ (v >> ns)) & fld;
        return v ^ (d << fs) ^ (d << ns);
      };
      hi.push_back({swap(m), swap(s), A->real_coeffs[(size_t)t]});
    }
  if (lo.empty() || hi.empty()) return 0;
  // the pieces of the packed operator sit one bit lower
  const int np = n - packed, fp = f - packed;
  dnm_subspace ld{}, rd{};
  ld.type = lh.type; ld.L = lh.L; ld.space = lh.space; ld.vec_swizzle = lh.swz;
  rd.type = rh.type; rd.L = rh.L; rd.space = rh.space; rd.vec_swizzle = rh.swz;
  const dnm_partition part{A->rank, A->nranks};
  dnm_mat *made[2] = {nullptr, nullptr};
  for (int which = 0; which < 2; ++which) {
    std::vector<Term> &T = which ? hi : lo;
    std::stable_sort(T.begin(), T.end(), [](const Term &a, const Term &b) { return a.m < b.m; });
    std::vector<int64_t> masks, offs, signs;
    std::vector<double> coeffs;
    for (size_t i = 0; i < T.size(); ++i) {
      if (i == 0 || T[i].m != T[i - 1].m) { masks.push_back(T[i].m); offs.push_back((int64_t)i); }
      signs.push_back(T[i].s);
      coeffs.push_back(T[i].c); coeffs.push_back(0.0);  // (one double per term is all a handle keeps: see dnm_mat_create)
    }
    offs.push_back((int64_t)T.size());
    int fl = A->flags & ~(0xff << DNM_MAT_AMIN_SHIFT);
    if (which == 1 && np - fp <= 8) {
      // layout B's masks live on the p exchanged bits and the top bit: a tile [0, a) + [f, n) holds them all and
      // leaves the bits right below f -- the top bits INSIDE a piece -- to the workgroup index, so that ranges of
      // workgroups are contiguous sub-pieces (dnm_mat_mult_local_part)
      const char *tb = knob("DNM_TILE_BITS");
      const int a = (tb ? atoi(tb) : 12) - (np - fp);
      if (a >= 2 && a <= 9) fl |= a << DNM_MAT_AMIN_SHIFT;
    } else {
      fl |= A->flags & (0xff << DNM_MAT_AMIN_SHIFT);
    }
    int rc = dnm_mat_create((int64_t)masks.size(), masks.data(), offs.data(), signs.data(), coeffs.data(), &ld, &rd, 0, fl,
                            &part, &made[which]);
    int ns_ = 0, nr_ = 0;
    if (rc == 0) rc = dnm_mat_exchange_plan(made[which], &ns_, nullptr, &nr_, nullptr);
    if (rc == 0 && (ns_ || nr_)) { set_error("transposed exchange: a pass is not rank-local"); rc = 1; }
    if (rc == 0 && !(made[which]->hypercube && made[which]->plan.use_tiled)) {
      set_error("transposed exchange: a part of the operator has no tiled plan"); rc = 1;
    }
    if (rc != 0) { delete made[0]; delete made[1]; return rc; }
  }
  A->tr_lo = made[0];
  A->tr_hi = made[1];
  A->tr_f = fp;
  *chosen = DNM_EXCHANGE_TRANSPOSE;
  return 0;
}

int dnm_mat_operator(const dnm_mat *A, int64_t *nmasks, int64_t *nterms, int64_t *masks, int64_t *mask_offsets,
                     int64_t *signs, double *coeffs) {
  DNM_CHECK(A && nmasks && nterms, "null argument");
  *nmasks = (int64_t)A->masks.size();
  *nterms = (int64_t)A->signs.size();
  if (masks) std::copy(A->masks.begin(), A->masks.end(), masks);
  if (mask_offsets) std::copy(A->mask_offsets.begin(), A->mask_offsets.end(), mask_offsets);
  if (signs) std::copy(A->signs.begin(), A->signs.end(), signs);
  if (coeffs) std::copy(A->real_coeffs.begin(), A->real_coeffs.end(), coeffs);
  return 0;
}

int dnm_mat_exchange_parts(const dnm_mat *A, dnm_mat **lo, dnm_mat **hi, int *f) {
  DNM_CHECK(A && lo && hi && f, "null argument");
  *lo = A->tr_lo; *hi = A->tr_hi; *f = A->tr_f;
  return 0;
}

int dnm_mat_sizes(const dnm_mat *A, int64_t *M, int64_t *N, int64_t *m_local, int64_t *n_local) {
  DNM_CHECK(A, "null matrix");
  if (M) *M = A->M;
  if (N) *N = A->N;
  if (m_local) *m_local = A->m_local;
  if (n_local) *n_local = A->n_local;
  return 0;
}

int dnm_mat_layouts(const dnm_mat *A, int *left, int *right) {
  DNM_CHECK(A, "null matrix");
  if (left) *left = A->use_sc3 ? A->left.host.sc3 : A->left.host.swz;
  if (right) *right = A->use_sc3 ? A->right.host.sc3 : A->right.host.swz;
  return 0;
}

int dnm_mat_is_real_packed(const dnm_mat *A, int *packed) {
  DNM_CHECK(A && packed, "null argument");
  *packed = A->real_packed ? 1 : 0;
  return 0;
}

// y = A x (- b z + c z2) in the SpinConserve internal layout; dot3 != null: the fused sums (device partials reduced here)
static int sc3_mult(dnm_mat *A, const void *x, void *y, const void *z, double b, const void *z2, double c_re, double c_im,
                    double *dot3_host, void *stream, int64_t win_start = -1, int phase = 0) {
  Sc3Call call;
  call.row0 = A->sc3->row0;
  call.win_start = win_start >= 0 ? win_start : A->sc3->row0;     // one rank: x is the whole vector
  call.zinit = (const double2 *)z;
  call.zscale = b;
  call.zinit2 = (const double2 *)z2;
  call.z2re = c_re;
  call.z2im = c_im;
  const double *dg = A->have_diag ? (const double *)A->diag.p : nullptr;
  if (!dot3_host) return launch_sc3(*A->sc3, A->dmsc_sc3, call, dg, x, y, S(stream), phase);
  const size_t nwg = sc3_dot_partials(*A->sc3);
  double *part = nullptr;
  DNM_TRY(vec_scratch(((nwg + 1) * 3 + vk_reduce_scratch(3)) * sizeof(double), &part));
  call.dot_out = part;
  DNM_TRY(launch_sc3(*A->sc3, A->dmsc_sc3, call, dg, x, y, S(stream)));
  DNM_TRY(vk_reduce_partials(part, (int)nwg, 3, part + 3 * nwg, S(stream), part + 3 * nwg + 3));
  DNM_HIP(hipMemcpyAsync(dot3_host, part + 3 * nwg, 3 * sizeof(double), hipMemcpyDeviceToHost, S(stream)));
  DNM_HIP(hipStreamSynchronize(S(stream)));
  return 0;
}

int dnm_mat_precompute_diagonal(dnm_mat *A, void *stream) {
  DNM_CHECK(A && !A->host_only, "null or host-only matrix");
  DNM_CHECK(!A->real_packed || (A->use_sc3 && A->sc3->diag_mode == 1),
            "this real-packed operator evaluates its diagonal on the fly: nothing to precompute");
  // only when the first mask is the identity (bpetsc_template_1.c:177-180) and
  // left == right (operators.py:627-629; the caller guarantees it)
  if (A->masks.empty() || A->masks[0] != 0) return 0;
  DNM_CHECK(A->nranks == 1 || !(A->hypercube && A->plan.use_tiled),
            "precomputed diagonal is not used by the partitioned tiled multiply");
  DNM_CHECK(A->M == A->N, "precompute_diagonal needs a square matrix");
  if (A->use_sc3) {
    // computed row by row in reference order, kept in the vectors' layout
    DevBuf nat;
    DNM_TRY(nat.alloc((size_t)A->rows_local * sizeof(double)));
    DNM_TRY(launch_diag(A->dmsc, A->right.dev, A->rows_local, A->row0, (double *)nat.p, S(stream)));
    // (one double per position of the layout; a real-packed handle counts its vectors in pairs of positions)
    DNM_TRY(A->diag.alloc((size_t)A->m_local * (A->real_packed ? 2 : 1) * sizeof(double)));
    DNM_TRY(sc3_layout_copy_f64(*A->sc3->ly, (double *)A->diag.p, (const double *)nat.p, true, S(stream), A->sc3->T0,
                                A->sc3->T1, &A->sc3->perm));
    DNM_HIP(hipStreamSynchronize(S(stream)));      // `nat` is released on return
    A->have_diag = true;
    return 0;
  }
  DNM_TRY(A->diag.alloc((size_t)A->m_local * sizeof(double)));
  DNM_TRY(launch_diag(A->dmsc, A->right.dev, A->m_local, A->row0, (double *)A->diag.p, S(stream)));
  A->have_diag = true;
  return 0;
}

int dnm_mat_get_diagonal(dnm_mat *A, double *diag_host, void *stream) {
  DNM_CHECK(A && A->have_diag, "no precomputed diagonal");
  if (A->use_sc3) {       // handed out in reference order (row r of the matrix)
    DevBuf nat;
    DNM_TRY(nat.alloc((size_t)A->rows_local * sizeof(double)));
    DNM_TRY(sc3_layout_copy_f64(*A->sc3->ly, (double *)nat.p, (const double *)A->diag.p, false, S(stream), A->sc3->T0,
                                A->sc3->T1, &A->sc3->perm));
    return dnm_memcpy_d2h(diag_host, nat.p, (size_t)A->rows_local * sizeof(double), stream);
  }
  return dnm_memcpy_d2h(diag_host, A->diag.p, (size_t)A->m_local * sizeof(double), stream);
}

// SpinConserve block kernel: the launch covers the high parts of the rows this rank owns.
static int launch_sc(dnm_mat *A, int64_t win_start, int64_t win_len, const void *xw, void *y, void *stream) {
  const double *dg = A->have_diag ? (const double *)A->diag.p : nullptr;
  if (A->scblock.lb)
    return launch_sc_block(A->dmsc, (const ScMask *)A->d_scmasks.p, A->scblock, A->right.dev, A->m_local, A->row0,
                           win_start, win_len, dg, xw, y, S(stream));
  return launch_sc_matvec(A->dmsc, (const ScMask *)A->d_scmasks.p, A->sclow, A->right.dev, A->m_local, A->row0,
                          win_start, dg, xw, y, nullptr, S(stream));
}



int dnm_mat_mult_local(dnm_mat *A, const void *x, void *y, void *stream) {
  DNM_CHECK(A && x && y, "null argument");
  DNM_CHECK(!A->host_only, "host-only handle cannot multiply");
  DNM_CHECK(x != y, "x and y must be different vectors");
  if (A->hypercube && A->plan.use_tiled) {
    for (auto &p : A->local_passes)
      DNM_TRY(launch_pass(A, *p, p->desc, x, y, nullptr, S(stream)));
    return 0;
  }
  DNM_CHECK(A->nranks == 1, "this subspace pair cannot run partitioned (use dnm_mat_mult_window)");
  if (A->use_sc3) return sc3_mult(A, x, y, nullptr, 0.0, nullptr, 0.0, 0.0, nullptr, stream);
  if (A->sc_pair) return launch_sc(A, 0, A->N, x, y, stream);
  return launch_gather_matvec(A->dmsc, A->left.dev, A->right.dev, A->M,
                              A->have_diag ? (const double *)A->diag.p : nullptr, x, y, S(stream));
}

// The rank-local passes over ONE of nparts equal ranges of their workgroups (block ids [part, part + 1) * grid / nparts):
// with an LDS-only plan whose tile holds the top index bits, range `part` reads and writes exactly the amplitudes whose
// highest non-tile index bits equal `part` -- the transposed exchange runs its layout-B pass sub-piece by sub-piece.
int dnm_mat_mult_local_part(dnm_mat *A, const void *x, void *y, int part, int nparts, void *stream) {
  DNM_CHECK(A && x && y && x != y && !A->host_only, "bad argument");
  DNM_CHECK(A->hypercube && A->plan.use_tiled, "only the tiled hypercube passes run over a range of their workgroups");
  DNM_CHECK(nparts >= 1 && (nparts & (nparts - 1)) == 0 && part >= 0 && part < nparts, "bad range %d of %d", part, nparts);
  for (auto &p : A->local_passes) {
    const unsigned grid = 1u << (p->n_eff - p->desc.tile_bits);
    DNM_CHECK((unsigned)nparts <= grid, "more ranges than workgroups");
    DevPass d = p->desc;
    d.block_offset = (uint32_t)part * (grid / (unsigned)nparts);
    DNM_TRY(launch_pass(A, *p, d, x, y, nullptr, S(stream), (unsigned)nparts));
  }
  return 0;
}

// what the top non-tile index bits of the rank-local passes are: *top_free_bit = the highest index bit that is in no
// pass's tile (the ranges of dnm_mat_mult_local_part split along it and the ones below it), -1 if passes disagree
int dnm_mat_local_part_bits(const dnm_mat *A, int *top_free_bit, int *gathers) {
  DNM_CHECK(A && top_free_bit && gathers, "null argument");
  *top_free_bit = -1;
  *gathers = 0;
  if (!(A->hypercube && A->plan.use_tiled)) return 0;
  int top = -2;
  for (auto &p : A->local_passes) {
    uint64_t tb = 0;
    for (int j = 0; j < p->desc.nseg; ++j) tb |= ((((uint64_t)1 << p->desc.seg_len[j]) - 1) << p->desc.seg_pos[j]);
    int hi = p->n_eff - 1;
    while (hi >= 0 && ((tb >> hi) & 1)) --hi;
    if (top == -2) top = hi; else if (top != hi) top = -1;
    *gathers += (int)(p->desc.loop[LP_COUNT] - p->desc.loop[LP_GATHER_REAL]);     // gathered records
    *gathers += (int)(p->desc.tab_loop[2] - p->desc.tab_loop[1]);
  }
  *top_free_bit = top < 0 ? -1 : top;
  return 0;
}

// y = A x - b z, <x, y> = sum conj(x_i) y_i and |y|^2 (z may be null).  When the plan is tiled, the first pass
// starts its accumulators from -b z instead of zero and the last pass -- if it stages x in LDS -- accumulates the
// sums (per-workgroup partials, summed by a second tiny kernel); otherwise one fused sweep does the same.
int dnm_mat_mult_lanczos(dnm_mat *A, const void *x, void *y, const void *z, double b, double *dot, void *stream) {
  DNM_CHECK(A && x && y && dot, "null argument");
  DNM_CHECK(A->remote_passes.empty(), "operator couples different ranks: no fused Lanczos step");
  DNM_CHECK(z != y && x != y, "y must not alias x or z");
  const bool tiled = A->hypercube && A->plan.use_tiled && !A->local_passes.empty() && !A->host_only;
  if (A->use_sc3 && !A->host_only) {
    if (A->sc3->tiled) return sc3_mult(A, x, y, z, b, nullptr, 0.0, 0.0, dot, stream);
    DNM_TRY(sc3_mult(A, x, y, z, b, nullptr, 0.0, 0.0, nullptr, stream));       // the row kernel takes the start vector
    return vec_lanczos_dot_host(y, nullptr, x, A->m_local, 0.0, dot, S(stream));
  }
  if (!tiled && A->sc_pair && A->scblock.lb && A->nranks == 1 && !A->host_only) {
    // SpinConserve block kernel: the beta term starts the accumulators, the sums are taken while the block of x
    // is still in LDS
    const size_t nwg = (size_t)sc_block_grid(A->scblock);
    double *part = nullptr;
    DNM_TRY(vec_scratch((nwg + 1) * 3 * sizeof(double), &part));
    DNM_TRY(launch_sc_block(A->dmsc, (const ScMask *)A->d_scmasks.p, A->scblock, A->right.dev, A->m_local, A->row0, 0,
                            A->N, A->have_diag ? (const double *)A->diag.p : nullptr, x, y, S(stream), z, b, part));
    DNM_TRY(vk_reduce_partials(part, (int)nwg, 3, part + 3 * nwg, S(stream)));
    DNM_HIP(hipMemcpyAsync(dot, part + 3 * nwg, 3 * sizeof(double), hipMemcpyDeviceToHost, S(stream)));
    DNM_HIP(hipStreamSynchronize(S(stream)));
    return 0;
  }
  if (!tiled) {
    DNM_TRY(dnm_mat_mult_local(A, x, y, stream));
    return vec_lanczos_dot_host(y, z, x, A->m_local, b, dot, S(stream));
  }
  const bool fused_dot = A->local_passes.back()->desc.need_tile != 0;
  const size_t nblk = pass_dot_partials(A, *A->local_passes.back());
  double *part = nullptr;
  if (fused_dot) DNM_TRY(vec_scratch(((nblk + 1) * 3 + vk_reduce_scratch(3)) * sizeof(double), &part));
  for (size_t i = 0; i < A->local_passes.size(); ++i) {
    DevPass d = A->local_passes[i]->desc;
    if (i == 0 && z) {
      DNM_CHECK(!d.accumulate, "internal: first pass accumulates");
      d.zinit = z;
      d.zscale = b;
    }
    if (fused_dot && i + 1 == A->local_passes.size()) d.dot_out = part;
    DNM_TRY(launch_pass(A, *A->local_passes[i], d, x, y, nullptr, S(stream)));
  }
  if (!fused_dot) return vec_lanczos_dot_host(y, nullptr, x, A->m_local, 0.0, dot, S(stream));
  DNM_TRY(vk_reduce_partials(part, (int)nblk, 3, part + 3 * nblk, S(stream), part + 3 * nblk + 3));
  DNM_HIP(hipMemcpyAsync(dot, part + 3 * nblk, 3 * sizeof(double), hipMemcpyDeviceToHost, S(stream)));
  DNM_HIP(hipStreamSynchronize(S(stream)));
  return 0;
}

// true when the multiply can start its accumulators from other vectors (dnm_mat_mult_sub2 costs no extra sweep)
int dnm_mat_fuses_init(const dnm_mat *A) {
  if (!A || A->host_only || !A->remote_passes.empty()) return 0;
  if (A->hypercube && A->plan.use_tiled && !A->local_passes.empty()) return 1;
  if (A->use_sc3) return 1;
  return (A->sc_pair && A->scblock.lb && A->nranks == 1) ? 1 : 0;
}

// y = A x - b z + c z2 without the sums (Chebyshev / Clenshaw recurrences; z2 may be null): the extra terms ride
// on the multiply where a kernel can start its accumulators from them, otherwise one more sweep each.
int dnm_mat_mult_sub2(dnm_mat *A, const void *x, void *y, const void *z, double b, const void *z2, double c_re,
                      double c_im, void *stream) {
  DNM_CHECK(A && x && y && z, "null argument");
  DNM_CHECK(A->remote_passes.empty(), "operator couples different ranks: use the partitioned multiply");
  DNM_CHECK(z != y && x != y && z2 != y, "y must not alias x, z or z2");
  DNM_CHECK(!A->host_only, "host-only handle cannot multiply");
  if (A->hypercube && A->plan.use_tiled && !A->local_passes.empty()) {
    for (size_t i = 0; i < A->local_passes.size(); ++i) {
      DevPass d = A->local_passes[i]->desc;
      if (i == 0) {
        DNM_CHECK(!d.accumulate, "internal: first pass accumulates");
        d.zinit = z;
        d.zscale = b;
        d.zinit2 = z2;
        d.z2re = c_re;
        d.z2im = c_im;
      }
      DNM_TRY(launch_pass(A, *A->local_passes[i], d, x, y, nullptr, S(stream)));
    }
    return 0;
  }
  if (A->use_sc3) return sc3_mult(A, x, y, z, b, z2, c_re, c_im, nullptr, stream);
  if (A->sc_pair && A->scblock.lb && A->nranks == 1)
    return launch_sc_block(A->dmsc, (const ScMask *)A->d_scmasks.p, A->scblock, A->right.dev, A->m_local, A->row0, 0,
                           A->N, A->have_diag ? (const double *)A->diag.p : nullptr, x, y, S(stream), z, b, nullptr,
                           z2, c_re, c_im);
  DNM_TRY(dnm_mat_mult_local(A, x, y, stream));
  DNM_TRY(vk_axpby(y, z, A->m_local, -b, 0.0, 1.0, 0.0, S(stream)));
  if (z2) DNM_TRY(vk_axpby(y, z2, A->m_local, c_re, c_im, 1.0, 0.0, S(stream)));
  return 0;
}

int dnm_mat_mult_sub(dnm_mat *A, const void *x, void *y, const void *z, double b, void *stream) {
  return dnm_mat_mult_sub2(A, x, y, z, b, nullptr, 0.0, 0.0, stream);
}

int dnm_mat_mult_dot(dnm_mat *A, const void *x, void *y, double *dot, void *stream) {
  DNM_CHECK(dot, "null argument");
  double d3[3];
  DNM_TRY(dnm_mat_mult_lanczos(A, x, y, nullptr, 0.0, d3, stream));
  dot[0] = d3[0];
  dot[1] = d3[1];
  return 0;
}

int dnm_mat_mult(dnm_mat *A, const void *x, void *y, void *stream) {
  DNM_CHECK(A, "null matrix");
  DNM_CHECK(A->remote_passes.empty(),
            "operator couples different ranks: use dnm_mat_mult_local + dnm_mat_mult_remote");
  return dnm_mat_mult_local(A, x, y, stream);
}

int dnm_mat_ownership(const dnm_mat *A, int64_t *row0, int64_t *m_local) {
  DNM_CHECK(A, "null matrix");
  // (internal SpinConserve layout: the rank's range of the layout -- the index space its windows are expressed in)
  if (row0) *row0 = A->use_sc3 ? (A->real_packed ? A->sc3->row0 / 2 : A->sc3->row0) : A->row0;
  if (m_local) *m_local = A->m_local;
  return 0;
}

int dnm_mat_column_window(dnm_mat *A, int64_t *cmin, int64_t *cmax, void *stream) {
  DNM_CHECK(A && cmin && cmax, "bad argument");
  if (A->use_sc3) {          // positions of the internal layout, from the T blocks the rank's rows reach (host tables)
    A->sc3->window(cmin, cmax);
    if (A->real_packed) {     // whole blocks: [cmin, cmax + 1) is a range of pairs
      *cmin /= 2;
      *cmax = (*cmax + 1) / 2 - 1;
    }
    return 0;
  }
  DNM_CHECK(!A->host_only, "bad argument");
  DNM_CHECK(!(A->hypercube && A->plan.use_tiled), "Full/Parity partitions on 2^p ranks exchange partner blocks, not windows");
  if (A->win_max < A->win_min) {
    const int nb = A->sc_pair ? sc_num_blocks(A->m_local) : gather_num_blocks(A->m_local);
    DevBuf buf;
    DNM_TRY(buf.alloc((size_t)nb * 2 * sizeof(int64_t)));
    if (A->sc_pair)
      DNM_TRY(launch_sc_matvec(A->dmsc, (const ScMask *)A->d_scmasks.p, A->sclow, A->right.dev, A->m_local, A->row0,
                               0, nullptr, nullptr, nullptr, (int64_t *)buf.p, S(stream)));
    else
      DNM_TRY(launch_gather_matvec(A->dmsc, A->left.dev, A->right.dev, A->m_local, nullptr, nullptr, nullptr,
                                   S(stream), A->row0, 0, 0, (int64_t *)buf.p));
    std::vector<int64_t> h((size_t)nb * 2);
    DNM_TRY(dnm_memcpy_d2h(h.data(), buf.p, h.size() * sizeof(int64_t), stream));
    // (a SpinConserve pair always reads its own rows' columns; a general pair reads what its masks reach)
    int64_t lo = A->sc_pair ? A->row0 : INT64_MAX, hi = A->sc_pair ? A->row0 + A->m_local - 1 : INT64_MIN;
    for (int b = 0; b < nb; ++b) { lo = std::min(lo, h[2 * b]); hi = std::max(hi, h[2 * b + 1]); }
    if (hi < lo) lo = hi = 0;          // no matrix element at all in these rows
    A->win_min = lo;
    A->win_max = hi;
  }
  *cmin = A->win_min;
  *cmax = A->win_max;
  return 0;
}

int dnm_mat_column_ranges(dnm_mat *A, int64_t max_ranges, int64_t *ranges, int64_t *nranges) {
  DNM_CHECK(A && nranges && max_ranges >= 0 && (ranges || max_ranges == 0), "bad argument");
  *nranges = 0;
  if (!A->use_sc3) return 0;             // (other partitions: dnm_mat_column_chunks)
  const auto rg = A->sc3->ranges();
  *nranges = (int64_t)rg.size();
  const int64_t unit = A->real_packed ? 2 : 1;       // a real-packed handle counts pairs of positions (whole blocks: even)
  for (int64_t i = 0; i < *nranges && i < max_ranges; ++i) {
    ranges[2 * i] = rg[(size_t)i].first / unit;
    ranges[2 * i + 1] = rg[(size_t)i].second / unit;
  }
  return 0;
}

int dnm_mat_column_chunks(dnm_mat *A, int chunk_shift, uint8_t *map, int64_t nchunks, void *stream) {
  DNM_CHECK(A && map && chunk_shift >= 0 && chunk_shift < 62, "bad argument");
  int64_t lo, hi;
  DNM_TRY(dnm_mat_column_window(A, &lo, &hi, stream));
  const int64_t first = lo >> chunk_shift;
  DNM_CHECK(nchunks == (hi >> chunk_shift) - first + 1, "the window [%lld, %lld] has %lld chunks of 2^%d columns",
            (long long)lo, (long long)hi, (long long)((hi >> chunk_shift) - first + 1), chunk_shift);
  if (A->use_sc3) {
    A->sc3->chunks(chunk_shift + (A->real_packed ? 1 : 0), first, nchunks, map);      // (chunks of 2^shift elements)
    return 0;
  }
  const int nb = A->sc_pair ? sc_num_blocks(A->m_local) : gather_num_blocks(A->m_local);
  DevBuf range, dmap;
  DNM_TRY(range.alloc((size_t)nb * 2 * sizeof(int64_t)));
  DNM_TRY(dmap.alloc((size_t)nchunks));
  DNM_HIP(hipMemsetAsync(dmap.p, 0, (size_t)nchunks, S(stream)));
  ColMark mark;
  mark.map = (uint8_t *)dmap.p;
  mark.shift = chunk_shift;
  mark.first = first;
  if (A->sc_pair)
    DNM_TRY(launch_sc_matvec(A->dmsc, (const ScMask *)A->d_scmasks.p, A->sclow, A->right.dev, A->m_local, A->row0,
                             0, nullptr, nullptr, nullptr, (int64_t *)range.p, S(stream), mark));
  else
    DNM_TRY(launch_gather_matvec(A->dmsc, A->left.dev, A->right.dev, A->m_local, nullptr, nullptr, nullptr,
                                 S(stream), A->row0, 0, 0, (int64_t *)range.p, mark));
  DNM_TRY(dnm_memcpy_d2h(map, dmap.p, (size_t)nchunks, stream));
  return 0;
}

int dnm_mat_mult_window(dnm_mat *A, const void *x_window, int64_t win_start, int64_t win_len, void *y_local,
                        void *stream) {
  DNM_CHECK(A && x_window && y_local && !A->host_only, "bad argument");
  int64_t lo, hi;
  DNM_TRY(dnm_mat_column_window(A, &lo, &hi, stream));
  DNM_CHECK(win_start <= lo && win_start + win_len > hi,
            "window [%lld, %lld) does not cover the columns [%lld, %lld] this rank reads", (long long)win_start,
            (long long)(win_start + win_len), (long long)lo, (long long)hi);
  if (A->use_sc3)
    return sc3_mult(A, x_window, y_local, nullptr, 0.0, nullptr, 0.0, 0.0, nullptr, stream,
                    A->real_packed ? 2 * win_start : win_start);
  if (A->sc_pair) return launch_sc(A, win_start, win_len, x_window, y_local, stream);
  // any other subspace pair: one thread per row, columns read from the window (index order)
  return launch_gather_matvec(A->dmsc, A->left.dev, A->right.dev, A->m_local,
                              A->have_diag ? (const double *)A->diag.p : nullptr, x_window, y_local, S(stream),
                              A->row0, win_start, 0, nullptr);
}

// Rows of a window partition whose columns all lie in [col_lo, col_hi) -- the rank's own block of x: they can be
// multiplied while the rest of the window is on the links.  From the per-workgroup column ranges of the row kernels
// (one sweep): runs of consecutive workgroups that read nothing else, the largest `max_ranges` of them, none
// shorter than min_blocks workgroups (of 256 rows).  ranges: pairs [r0, r1) of local row numbers, ascending.
int dnm_mat_window_local_rows(dnm_mat *A, int64_t col_lo, int64_t col_hi, int max_ranges, int min_blocks,
                              int64_t *ranges, int *nranges, void *stream) {
  DNM_CHECK(A && ranges && nranges && max_ranges >= 1, "bad argument");
  *nranges = 0;
  if (A->use_sc3 || A->host_only || (A->hypercube && A->plan.use_tiled) || A->m_local <= 0) return 0;
  if (A->left.host.swz != 0) return 0;       // a swizzled result block is not a sequence of row ranges
  const int64_t per = A->sc_pair ? sc_rows_per_block() : gather_rows_per_block();
  const int nb = A->sc_pair ? sc_num_blocks(A->m_local) : gather_num_blocks(A->m_local);
  DevBuf buf;
  DNM_TRY(buf.alloc((size_t)nb * 2 * sizeof(int64_t)));
  if (A->sc_pair)
    DNM_TRY(launch_sc_matvec(A->dmsc, (const ScMask *)A->d_scmasks.p, A->sclow, A->right.dev, A->m_local, A->row0,
                             0, nullptr, nullptr, nullptr, (int64_t *)buf.p, S(stream)));
  else
    DNM_TRY(launch_gather_matvec(A->dmsc, A->left.dev, A->right.dev, A->m_local, nullptr, nullptr, nullptr,
                                 S(stream), A->row0, 0, 0, (int64_t *)buf.p));
  std::vector<int64_t> h((size_t)nb * 2);
  DNM_TRY(dnm_memcpy_d2h(h.data(), buf.p, h.size() * sizeof(int64_t), stream));
  // (a SpinConserve pair reads its own rows' columns as well: the diagonal and the row's own amplitude)
  struct Run { int64_t b0, b1; };
  std::vector<Run> runs;
  for (int b = 0; b < nb;) {
    auto local = [&](int i) {
      int64_t lo = h[2 * (size_t)i], hi = h[2 * (size_t)i + 1];
      if (A->sc_pair) {
        lo = std::min(lo, A->row0 + (int64_t)i * per);
        hi = std::max(hi, std::min(A->row0 + A->m_local, A->row0 + (int64_t)(i + 1) * per) - 1);
      }
      return hi < lo || (lo >= col_lo && hi < col_hi);
    };
    if (!local(b)) { ++b; continue; }
    int e = b;
    while (e < nb && local(e)) ++e;
    if (e - b >= std::max(1, min_blocks)) runs.push_back({b, e});
    b = e;
  }
  std::sort(runs.begin(), runs.end(), [](const Run &a, const Run &b) { return a.b1 - a.b0 > b.b1 - b.b0; });
  if ((int)runs.size() > max_ranges) runs.resize((size_t)max_ranges);
  std::sort(runs.begin(), runs.end(), [](const Run &a, const Run &b) { return a.b0 < b.b0; });
  for (const Run &r : runs) {
    ranges[2 * *nranges] = r.b0 * per;
    ranges[2 * *nranges + 1] = std::min(A->m_local, r.b1 * per);
    ++*nranges;
  }
  return 0;
}

// dnm_mat_mult_window for the local rows [r0, r1) only (y_local is still the rank's whole result block)
int dnm_mat_mult_window_rows(dnm_mat *A, const void *x_window, int64_t win_start, int64_t win_len, void *y_local,
                             int64_t r0, int64_t r1, void *stream) {
  DNM_CHECK(A && x_window && y_local && !A->host_only, "bad argument");
  DNM_CHECK(!A->use_sc3 && !(A->hypercube && A->plan.use_tiled), "this operator does not multiply by row ranges");
  DNM_CHECK(r0 >= 0 && r0 < r1 && r1 <= A->m_local, "row range [%lld, %lld) outside the %lld local rows", (long long)r0,
            (long long)r1, (long long)A->m_local);
  const double *dg = A->have_diag ? (const double *)A->diag.p + r0 : nullptr;
  char *y = (char *)y_local + (size_t)r0 * 16;
  if (A->sc_pair && A->scblock.lb) {
    ScBlock blk = A->scblock;
    blk.h_first = sub_i2s(A->row0 + r0, A->left.host) >> blk.lb;
    blk.h_last = sub_i2s(A->row0 + r1 - 1, A->left.host) >> blk.lb;
    blk.swizzle = 0;
    blk.perm = nullptr;
    blk.nperm = 0;
    return launch_sc_block(A->dmsc, (const ScMask *)A->d_scmasks.p, blk, A->right.dev, r1 - r0, A->row0 + r0, win_start,
                           win_len, dg, x_window, y, S(stream));
  }
  if (A->sc_pair)
    return launch_sc_matvec(A->dmsc, (const ScMask *)A->d_scmasks.p, A->sclow, A->right.dev, r1 - r0, A->row0 + r0,
                            win_start, dg, x_window, y, nullptr, S(stream));
  return launch_gather_matvec(A->dmsc, A->left.dev, A->right.dev, r1 - r0, dg, x_window, y, S(stream), A->row0 + r0,
                              win_start, 0, nullptr);
}

int dnm_mat_window_split(const dnm_mat *A, int *supported) {
  DNM_CHECK(A && supported, "null argument");
  *supported = (A->use_sc3 && A->sc3->tiled && !A->sc3->graph && A->nranks > 1) ? 1 : 0;    // (a bond graph's lo pass reads other blocks)
  return 0;
}

int dnm_mat_mult_window_local(dnm_mat *A, const void *x_local, void *y_local, void *stream) {
  DNM_CHECK(A && x_local && y_local && !A->host_only, "bad argument");
  DNM_CHECK(A->use_sc3 && A->sc3->tiled, "this operator's window multiply does not split (dnm_mat_window_split)");
  // the lo pass reads the rank's own rows and, for the Lo/W boundary bond, other rows of the same T block: all local
  return sc3_mult(A, x_local, y_local, nullptr, 0.0, nullptr, 0.0, 0.0, nullptr, stream, A->sc3->row0, 1);
}

int dnm_mat_mult_window_remote(dnm_mat *A, const void *x_window, int64_t win_start, int64_t win_len, void *y_local,
                               void *stream) {
  DNM_CHECK(A && x_window && y_local && !A->host_only, "bad argument");
  DNM_CHECK(A->use_sc3 && A->sc3->tiled, "this operator's window multiply does not split (dnm_mat_window_split)");
  int64_t lo, hi;
  DNM_TRY(dnm_mat_column_window(A, &lo, &hi, stream));
  DNM_CHECK(win_start <= lo && win_start + win_len > hi,
            "window [%lld, %lld) does not cover the positions [%lld, %lld] this rank reads", (long long)win_start,
            (long long)(win_start + win_len), (long long)lo, (long long)hi);
  return sc3_mult(A, x_window, y_local, nullptr, 0.0, nullptr, 0.0, 0.0, nullptr, stream,
                  A->real_packed ? 2 * win_start : win_start, 2);
}

int dnm_mat_exchange_plan(const dnm_mat *A, int *nsend, dnm_xfer *sends, int *nrecv, dnm_xfer *recvs) {
  DNM_CHECK(A && nsend && nrecv, "null argument");
  *nsend = (int)A->plan.sends.size();
  *nrecv = (int)A->remote_passes.size();
  if (sends)
    for (size_t i = 0; i < A->plan.sends.size(); ++i)
      sends[i] = {A->plan.sends[i].partner, -1, A->plan.sends[i].offset, A->plan.sends[i].count};
  if (recvs)
    for (size_t i = 0; i < A->remote_passes.size(); ++i) {
      const auto &p = A->remote_passes[i];
      recvs[i] = {p->partner, (int32_t)i, p->src_off, (int64_t)1 << p->n_eff};
    }
  return 0;
}

int dnm_mat_mult_remote(dnm_mat *A, int32_t recv_index, const void *x_recv, void *y, void *stream) {
  DNM_CHECK(A && x_recv && y, "null argument");
  DNM_CHECK(!A->host_only, "host-only handle cannot multiply");
  DNM_CHECK(recv_index >= 0 && recv_index < (int)A->remote_passes.size(), "rank %d has no receive %d", A->rank,
            recv_index);
  const auto &p = A->remote_passes[recv_index];
  return launch_pass(A, *p, p->desc, x_recv, (char *)y + (size_t)p->y_off * 16, x_recv, S(stream));
}

int dnm_mat_norm_inf(dnm_mat *A, double *nrm, void *stream) {
  DNM_CHECK(A && nrm, "null argument");
  DNM_CHECK(!A->host_only, "host-only handle has no device tables");
  if (A->nrm != -1.0) {
    *nrm = A->nrm;
    return 0;
  }
  const int nb = norm_num_blocks(A->rows_local);
  DNM_TRY(A->scratch.alloc((size_t)nb * sizeof(double)));
  DNM_TRY(launch_norm(A->dmsc, A->left.dev, A->right.dev, A->rows_local, A->row0,
                      (double *)A->scratch.p, S(stream)));
  std::vector<double> h(nb);
  DNM_TRY(dnm_memcpy_d2h(h.data(), A->scratch.p, (size_t)nb * sizeof(double), stream));
  double best = 0.0;
  for (double v : h) best = std::max(best, v);
  if (A->nranks == 1) A->nrm = best;   // partitioned: caller reduces, then dnm_mat_set_norm
  *nrm = best;
  return 0;
}

int dnm_mat_set_norm(dnm_mat *A, double nrm) {
  DNM_CHECK(A, "null matrix");
  A->nrm = nrm;
  return 0;
}

int dnm_mat_plan_describe(const dnm_mat *A, char *buf, size_t buflen) {
  DNM_CHECK(A && buf && buflen, "null argument");
  std::string s;
  if (A->hypercube) s = A->plan.describe(A->op);
  else if (A->use_sc3) {
    const Sc3Tab &T = A->sc3->ly->host;
    char tmp[512];
    if (A->sc3->tiled && A->sc3->graph)
      snprintf(tmp, sizeof tmp, "SpinConserve two-pass kernels on a bond graph, internal layout [T %d | W %d | Lo %d]: "
               "window pass (%zu workgroups, %d hops in LDS, %d gathered) then lo pass (%zu workgroups, %d hops in LDS, "
               "%d gathered), diagonal %s, coefficients %s\n", T.t, T.w, T.a, A->sc3->permB.size(), A->sc3->op.nldsB,
               A->sc3->op.ngatB, A->sc3->permA.size() / 8, A->sc3->op.nldsA, A->sc3->op.ngatA,
               A->sc3->diag_mode == 2 ? "on the fly" : (A->sc3->diag_mode == 1 ? "cached" : "none"),
               A->sc3->sym ? "real symmetric" : "complex");
    else if (A->sc3->tiled)
      snprintf(tmp, sizeof tmp, "SpinConserve two-pass kernels, internal layout [T %d | W %d | Lo %d]: window pass (%zu "
               "workgroups, %d bonds in LDS, %d gathered) then lo pass (%zu workgroups, %d bonds in LDS, %d gathered), "
               "diagonal %s, coefficients %s\n", T.t, T.w, T.a, A->sc3->permB.size(), T.w - 1,
               __builtin_popcountll(A->sc3->op.bondsB), A->sc3->permA.size() / 8, T.a - 1,
               __builtin_popcountll(A->sc3->op.bondsA),
               A->sc3->diag_mode == 2 ? "on the fly" : (A->sc3->diag_mode == 1 ? "cached" : "none"),
               A->sc3->sym ? "real symmetric" : "complex");
    else
      snprintf(tmp, sizeof tmp, "SpinConserve row kernel, internal layout [T %d | W %d | Lo %d] (positions by table)\n",
               T.t, T.w, T.a);
    s = tmp;
  } else if (A->sc_pair)
    s = A->scblock.lb ? "SpinConserve kernel, block form (" + std::to_string(A->scblock.lb) +
                            " low bits per workgroup in LDS, high bonds as block runs)\n"
                      : std::string("SpinConserve kernel (one row per thread, incremental colex rank)\n");
  else s = A->nranks > 1 ? "generic row-gather kernel (rows split in index order, columns through a window)\n"
                         : "generic row-gather kernel (non-hypercube subspace pair)\n";
  if (A->hypercube && !A->plan.use_tiled) s += "generic row-gather kernel in use\n";
  if (A->hypercube && A->plan.use_tiled) {
    // masks of many terms run as table records (plan.h: DevTab); said only when there are any: the plans without keep their text
    size_t ntab = 0, nquad = 0;
    for (const auto &p : A->local_passes) { ntab += p->h_tabs.size(); nquad += p->desc.loop[LP_COUNT] - p->desc.loop[0]; }
    for (const auto &p : A->remote_passes) { ntab += p->h_tabs.size(); nquad += p->desc.loop[LP_COUNT] - p->desc.loop[0]; }
    if (ntab) s += "table records: " + std::to_string(ntab) + " (beside " + std::to_string(nquad) + " records of four terms)\n";
  }
  snprintf(buf, buflen, "%s", s.c_str());
  return 0;
}

int dnm_mat_export_pass(const dnm_mat *A, int remote, int idx, void *desc_out, size_t desc_bytes,
                        void *quads_out, size_t quad_bytes, int max_quads, int *nquads) {
  DNM_CHECK(A && nquads, "null argument");
  const auto &v = remote ? A->remote_passes : A->local_passes;
  DNM_CHECK(idx >= 0 && idx < (int)v.size(), "pass index out of range");
  const PassOnDevice &p = *v[idx];
  *nquads = (int)p.h_quads.size();
  if (desc_out) {
    DNM_CHECK(desc_bytes == sizeof(DevPass), "DevPass size mismatch (%zu vs %zu)", desc_bytes, sizeof(DevPass));
    memcpy(desc_out, &p.desc, sizeof(DevPass));
  }
  if (quads_out) {
    DNM_CHECK(quad_bytes == sizeof(DevQuad), "DevQuad size mismatch (%zu vs %zu)", quad_bytes, sizeof(DevQuad));
    DNM_CHECK(max_quads >= *nquads, "record buffer too small");
    if (!p.h_quads.empty()) memcpy(quads_out, p.h_quads.data(), p.h_quads.size() * sizeof(DevQuad));   // (an empty pass: no null source)
  }
  return 0;
}

int dnm_mat_export_tabs(const dnm_mat *A, int remote, int idx, void *tabs_out, size_t tab_bytes, int max_tabs, int *ntabs,
                        double *vals_out, int64_t max_vals, int64_t *nvals) {
  DNM_CHECK(A && ntabs && nvals, "null argument");
  const auto &v = remote ? A->remote_passes : A->local_passes;
  DNM_CHECK(idx >= 0 && idx < (int)v.size(), "pass index out of range");
  const PassOnDevice &p = *v[idx];
  *ntabs = (int)p.h_tabs.size();
  *nvals = (int64_t)p.h_tabvals.size();
  if (tabs_out && !p.h_tabs.empty()) {
    DNM_CHECK(tab_bytes == sizeof(DevTab), "DevTab size mismatch (%zu vs %zu)", tab_bytes, sizeof(DevTab));
    DNM_CHECK(max_tabs >= *ntabs, "record buffer too small");
    memcpy(tabs_out, p.h_tabs.data(), p.h_tabs.size() * sizeof(DevTab));
  }
  if (vals_out && !p.h_tabvals.empty()) {
    DNM_CHECK(max_vals >= *nvals, "table buffer too small");
    memcpy(vals_out, p.h_tabvals.data(), p.h_tabvals.size() * sizeof(double));
  }
  return 0;
}

int dnm_mat_export_dtile(const dnm_mat *A, int remote, int idx, double *out, int64_t n) {
  DNM_CHECK(A && out, "null argument");
  const auto &v = remote ? A->remote_passes : A->local_passes;
  DNM_CHECK(idx >= 0 && idx < (int)v.size(), "pass index out of range");
  const PassOnDevice &p = *v[idx];
  DNM_CHECK((int64_t)p.h_dtile.size() == n, "the pass has %zu tabulated diagonal entries, not %lld", p.h_dtile.size(),
            (long long)n);
  memcpy(out, p.h_dtile.data(), p.h_dtile.size() * sizeof(double));
  return 0;
}

int dnm_mat_plan_counts(const dnm_mat *A, int *n_local_passes, int *n_remote_passes, int *tiled,
                        int *B, int *logR, int *n_loc) {
  DNM_CHECK(A, "null matrix");
  if (n_local_passes) *n_local_passes = (int)A->local_passes.size();
  if (n_remote_passes) *n_remote_passes = (int)A->remote_passes.size();
  if (tiled) *tiled = (A->hypercube && A->plan.use_tiled) ? 1 : 0;
  if (B) *B = A->plan.cfg.B;
  if (logR) *logR = A->plan.cfg.logR;
  if (n_loc) *n_loc = A->plan.n_loc;
  return 0;
}

int dnm_mat_plan_launches(const dnm_mat *A, int *n) {
  DNM_CHECK(A && n, "null argument");
  *n = (A->hypercube && A->plan.use_tiled) ? (int)A->local_passes.size() : 1;
  return 0;
}

}  // extern "C"

dnm_mat::~dnm_mat() {
  delete tr_lo;
  delete tr_hi;
}
