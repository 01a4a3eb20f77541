// Host-side planner: decides which masks each pass of the tiled kernel serves
// from its LDS tile and which it gathers from global memory.
#include "plan.h"

#include <algorithm>
#include <cstdlib>
#include <sstream>

namespace dnm {

static int env_int(const char *name, int dflt) {
  const char *v = knob(name);
  if (!v) return dflt;
  return atoi(v);
}

PlanConfig plan_config_from_env() {
  PlanConfig c;
  c.B = env_int("DNM_TILE_BITS", c.B);
  c.logR = env_int("DNM_LOG_ROWS", c.logR);
  c.amin = env_int("DNM_AMIN", c.amin);
  c.mode = env_int("DNM_PLAN_MODE", c.mode);
  c.gbits = env_int("DNM_GBITS", c.gbits);
  c.gbits_window = env_int("DNM_GBITS_WINDOW", c.gbits_window);
  c.window_first = env_int("DNM_WINDOW_FIRST", c.window_first);
  c.Bw = env_int("DNM_TILE_BITS_WINDOW", c.Bw);
  c.logRw = env_int("DNM_LOG_ROWS_WINDOW", c.logRw);
  c.cache_policy = env_int("DNM_CACHE_POLICY", c.cache_policy);
  if (const char *d = knob("DNM_DIAG_PASS")) c.diag_last = (d[0] == 'l') ? 1 : 0;
  return c;
}

static inline uint64_t lowmask(int nbits) {
  return nbits >= 64 ? ~0ull : (((uint64_t)1 << nbits) - 1);
}

// Tile of B bits: low segment [0,a) plus window [w, w+B-a) (a==B: low only).
static PassSpec tile_spec(int B, int a, int w) {
  PassSpec p;
  p.B = B;
  if (a >= B) {
    p.nseg = 1;
    p.seg_len[0] = B;
    p.seg_pos[0] = 0;
  } else if (w == a) {        // contiguous after all
    p.nseg = 1;
    p.seg_len[0] = B;
    p.seg_pos[0] = 0;
  } else {
    p.nseg = 2;
    p.seg_len[0] = a;
    p.seg_pos[0] = 0;
    p.seg_len[1] = B - a;
    p.seg_pos[1] = w;
  }
  return p;
}

// ---- partner exchange ----------------------------------------------------------
// A mask that flips rank bits couples this rank's rows to the block of rank
// `rank ^ (mask >> nl)`.  Its matrix elements are
//     c(row) = sum_t coeff_t (-1)^popcount(row & sign_t),
// a combination of Walsh functions of the row bits, which vanishes on a sub-block
// (rank bits and the top local bit fixed) iff the coefficients of every distinct
// Walsh function of the remaining bits cancel.  Flip-flop terms (XX+YY) vanish on
// half of the sub-blocks, so half of the traffic and passes disappear.
struct Need {
  int partner;
  int n_eff;
  int64_t y_off, src_off;
  std::vector<int> masks;
};

static bool vanishes_on(const RowMask &m, uint64_t fixed_bits, uint64_t fixed_value) {
  std::vector<std::pair<std::pair<uint64_t, int>, double>> groups;   // (free sign bits, is_imag) -> sum
  for (const RowTerm &t : m.terms) {
    const double c = (__builtin_popcountll(t.sign & fixed_value) & 1) ? -t.coeff : t.coeff;
    const std::pair<uint64_t, int> key{t.sign & ~fixed_bits, t.is_imag};
    bool found = false;
    for (auto &g : groups)
      if (g.first == key) { g.second += c; found = true; break; }
    if (!found) groups.push_back({key, c});
  }
  for (auto &g : groups) if (g.second != 0.0) return false;
  return true;
}

// Receives of rank `rank`, sorted by (partner, sub-block): the same function is
// evaluated for the partner ranks to derive the matching sends.
static void remote_needs(const OpForm &op, int rank, int nl, int B, std::vector<Need> *out) {
  out->clear();
  const int levels = (nl - 1 >= B) ? 1 : 0;          // split the block in halves when a half still tiles
  const int n_eff = nl - levels;
  const uint64_t highbits = ~lowmask(n_eff);
  for (int i = 0; i < (int)op.masks.size(); ++i) {
    const uint64_t m = op.masks[i].mask;
    const uint64_t h = m >> nl;
    if (h == 0) continue;
    const int partner = rank ^ (int)h;
    for (int half = 0; half < (1 << levels); ++half) {
      const uint64_t fixed = ((uint64_t)rank << nl) | ((uint64_t)half << n_eff);
      if (vanishes_on(op.masks[i], highbits, fixed)) continue;
      const int64_t y_off = (int64_t)half << n_eff;
      const int64_t src_off = (int64_t)((((uint64_t)half << n_eff) ^ m) & lowmask(nl) & highbits);
      Need *slot = nullptr;
      for (Need &nd : *out)
        if (nd.partner == partner && nd.y_off == y_off && nd.src_off == src_off) slot = &nd;
      if (!slot) {
        out->push_back({partner, n_eff, y_off, src_off, {}});
        slot = &out->back();
      }
      slot->masks.push_back(i);
    }
  }
  std::stable_sort(out->begin(), out->end(), [](const Need &a, const Need &b) {
    if (a.partner != b.partner) return a.partner < b.partner;
    if (a.y_off != b.y_off) return a.y_off < b.y_off;
    return a.src_off < b.src_off;
  });
}

static int make_plan_with(const OpForm &op, int rank, int nranks, const PlanConfig &cfg_in, Plan *out) {
  Plan &pl = *out;
  pl = Plan();
  pl.cfg = cfg_in;
  PlanConfig &cfg = pl.cfg;
  pl.n = op.n;
  pl.rank = rank;
  pl.nranks = nranks;
  DNM_CHECK(nranks >= 1 && (nranks & (nranks - 1)) == 0, "nranks must be a power of two");
  int p = ilog2((uint64_t)nranks);
  pl.n_loc = op.n - p;
  DNM_CHECK(pl.n_loc >= 1, "too many ranks for this dimension");
  DNM_CHECK(pl.n_loc <= 32, "local vector longer than 2^32 amplitudes is not supported");
  if (cfg.B < 8) cfg.B = 8;
  if (cfg.B > 13) cfg.B = 13;
  if (cfg.logR < 2) cfg.logR = 2;
  if (cfg.logR > 4) cfg.logR = 4;
  if (cfg.amin < 2) cfg.amin = 2;
  const int B = cfg.B;
  const int nl = pl.n_loc;
  const uint64_t locmask = lowmask(nl);

  // split masks into local (no rank bit flipped) and remote, by partner
  std::vector<int> local_masks;
  std::vector<std::pair<int, int>> remote;  // (partner, mask idx)
  bool has_diag = false;
  for (int i = 0; i < (int)op.masks.size(); ++i) {
    uint64_t m = op.masks[i].mask;
    uint64_t h = m >> nl;
    if (h == 0) {
      if (m == 0 && !op.masks[i].zero_mask_offdiag) has_diag = true; else local_masks.push_back(i);
    } else {
      remote.push_back({(int)(rank ^ (int)h), i});
    }
  }

  pl.use_tiled = nl >= B;
  if (!pl.use_tiled) return 0;

  std::vector<int> remaining = local_masks;
  auto covered_by = [&](const PassSpec &ps, std::vector<int> *cov) {
    uint64_t tb = ps.tile_bits();
    int c = 0;
    for (int idx : remaining)
      if ((op.masks[idx].mask & locmask & ~tb) == 0) {
        ++c;
        if (cov) cov->push_back(idx);
      }
    return c;
  };

  if (cfg.mode == 1) {
    // single pass: contiguous low tile, everything outside it gathered
    PassSpec ps = tile_spec(B, B, 0);
    std::vector<int> cov;
    covered_by(ps, &cov);
    ps.tile_masks = cov;
    for (int idx : remaining)
      if (std::find(cov.begin(), cov.end(), idx) == cov.end()) {
        ps.gather_masks.push_back(idx);
        ps.gather_src.push_back(0);
      }
    ps.has_diag = has_diag;
    pl.local.push_back(ps);
  } else if (cfg.mode == 2) {
    // greedy cover by (LDS tile, XCD group) pairs: masks inside the tile come
    // from LDS, masks that also touch the group bits are gathered (L2)
    // measured on MI355X (profiles/r01_sweep6.txt): with early gathers and streaming y traffic a wide group pays
    // from 2^26 local amplitudes on (9 bits at 2^30)
    // with swizzled vectors the window passes' gathers merge in the L2 like the contiguous pass's, so the group
    // stays at what an XCD's L2 holds (2^6 tiles of 64 KB): 83 B/amp against 115 (profiles/r02_exp1_swz.txt)
    if (cfg.gbits < 0) cfg.gbits = cfg.swz ? 6 : (nl >= 30 ? 9 : (nl >= 26 ? 8 : 6));
    if (cfg.gbits > 10) cfg.gbits = 10;
    bool first = true;
    while (!remaining.empty() || first) {
      PassSpec best;
      int best_score = -1, best_tile = -1;
      auto consider = [&](PassSpec ps) {
        const uint64_t tb = ps.tile_bits();
        int in_tile = 0;
        for (int idx : remaining)
          if ((op.masks[idx].mask & locmask & ~tb) == 0) ++in_tile;
        for (int glen = 0; glen <= cfg.gbits; ++glen) {
          for (int g = 0; g + glen <= nl; ++g) {
            const uint64_t gb = glen ? ((((uint64_t)1 << glen) - 1) << g) : 0;
            if (gb & tb) continue;
            int sc = 0;
            for (int idx : remaining)
              if ((op.masks[idx].mask & locmask & ~(tb | gb)) == 0) ++sc;
            // most masks covered; then most of them from LDS (fewest gathers);
            // then the smallest XCD group; earlier candidates win remaining ties
            const bool better = sc > best_score || (sc == best_score && in_tile > best_tile) ||
                                (sc == best_score && in_tile == best_tile && glen < best.glen);
            if (better) {
              best_score = sc;
              best_tile = in_tile;
              best = ps;
              best.glen = glen;
              best.gpos = g;
            }
            if (glen == 0) break;
          }
        }
      };
      consider(tile_spec(B, B, 0));
      const int Bw = (cfg.Bw >= 8 && cfg.Bw <= 13 && nl >= cfg.Bw) ? cfg.Bw : B;
      for (int a = cfg.amin; a < Bw && a <= 9; ++a)
        for (int w = a + 1; w + (Bw - a) <= nl; ++w) consider(tile_spec(Bw, a, w));
      if (best_score <= 0 && !first) break;
      const uint64_t tb = best.tile_bits();
      const uint64_t gb = best.glen ? ((((uint64_t)1 << best.glen) - 1) << best.gpos) : 0;
      std::vector<int> rest;
      for (int idx : remaining) {
        const uint64_t m = op.masks[idx].mask & locmask;
        if ((m & ~tb) == 0) best.tile_masks.push_back(idx);
        else if ((m & ~(tb | gb)) == 0) { best.gather_masks.push_back(idx); best.gather_src.push_back(0); }
        else rest.push_back(idx);
      }
      best.has_diag = first && has_diag;
      best.accumulate = !first;
      if (best.nseg > 1 && cfg.logRw > 0) best.logR = cfg.logRw;
      // Window tiles (runs of 2^a amplitudes far apart) alias in the L2 sets: only a few hundred KB of
      // them stay resident (tools/l2map_probe.hip).  Keep the masks, but order the workgroups by a
      // smaller subcube -- the top bits of the span -- so that at least those gathers find their lines.
      if (best.nseg > 1 && cfg.gbits_window >= 0 && best.glen > cfg.gbits_window) {
        best.gpos += best.glen - cfg.gbits_window;
        best.glen = cfg.gbits_window;
      }
      pl.local.push_back(best);
      remaining.swap(rest);
      first = false;
    }
    for (int idx : remaining) {
      pl.local[0].gather_masks.push_back(idx);
      pl.local[0].gather_src.push_back(0);
    }
  } else {
    // greedy cover by LDS tiles
    bool first = true;
    while (!remaining.empty() || first) {
      PassSpec best;
      int best_score = -1;
      // candidates: pure low tile, then low segment a + window
      {
        PassSpec ps = tile_spec(B, B, 0);
        int sc = covered_by(ps, nullptr);
        if (sc > best_score) { best_score = sc; best = ps; }
      }
      for (int a = cfg.amin; a < B && a <= 9; ++a) {
        int b = B - a;
        for (int w = a + 1; w + b <= nl; ++w) {
          PassSpec ps = tile_spec(B, a, w);
          int sc = covered_by(ps, nullptr);
          // prefer more coverage, then longer contiguous runs
          if (sc > best_score) { best_score = sc; best = ps; }
        }
      }
      if (best_score <= 0 && !first) break;   // nothing placeable: gather the rest
      std::vector<int> cov;
      covered_by(best, &cov);
      best.tile_masks = cov;
      best.has_diag = first && has_diag;
      best.accumulate = !first;
      pl.local.push_back(best);
      std::vector<int> rest;
      for (int idx : remaining)
        if (std::find(cov.begin(), cov.end(), idx) == cov.end()) rest.push_back(idx);
      remaining.swap(rest);
      first = false;
    }
    // masks no two-segment tile can hold: gathered in the first pass
    for (int idx : remaining) {
      pl.local[0].gather_masks.push_back(idx);
      pl.local[0].gather_src.push_back(0);
    }
  }

  // Execution order (DNM_WINDOW_FIRST; the default is set where the plan is requested, mat.cpp): window passes
  // first, the contiguous pass accumulating last -- the bandwidth-bound window pass then need not read y.
  if (cfg.window_first > 0 && pl.local.size() > 1) {
    std::stable_sort(pl.local.begin(), pl.local.end(),
                     [](const PassSpec &a, const PassSpec &b) { return (a.nseg > 1) > (b.nseg > 1); });
    for (size_t i = 0; i < pl.local.size(); ++i) {
      pl.local[i].accumulate = i > 0;
      pl.local[i].has_diag = i == 0 && has_diag;
    }
  }

  // the diagonal rides on the last, accumulating pass (DNM_DIAG_PASS=first|last; the default is set where the plan
  // is requested, mat.cpp): that pass is bound by its 48 B/amp, the first one by its records
  // (profiles/r02_exp13_diag.txt, r02_exp45_after_order.txt, r02_exp46_rows4_series.txt)
  if (cfg.diag_last > 0 && pl.local.size() > 1 && has_diag) {
    for (auto &ps : pl.local) ps.has_diag = false;
    pl.local.back().has_diag = true;
  }

  // remote passes: what this rank receives and applies
  std::vector<Need> mine;
  remote_needs(op, rank, nl, B, &mine);
  for (const Need &nd : mine) {
    PassSpec ps = tile_spec(B, B, 0);
    ps.accumulate = true;
    ps.partner = nd.partner;
    ps.n_eff = nd.n_eff;
    ps.y_off = nd.y_off;
    ps.src_off = nd.src_off;
    ps.sign_extra = (uint64_t)nd.y_off;
    for (int idx : nd.masks) {
      ps.gather_masks.push_back(idx);
      ps.gather_src.push_back(1);
    }
    pl.remote.push_back(ps);
  }
  // ... and what the partners receive from it: evaluate their needs the same way
  std::vector<int> peers;
  for (const auto &r : remote)
    if (std::find(peers.begin(), peers.end(), r.first) == peers.end()) peers.push_back(r.first);
  std::sort(peers.begin(), peers.end());
  for (int q : peers) {
    std::vector<Need> theirs;
    remote_needs(op, q, nl, B, &theirs);
    for (const Need &nd : theirs)
      if (nd.partner == rank) pl.sends.push_back({q, nd.src_off, (int64_t)1 << nd.n_eff});
  }
  return 0;
}

int make_plan(const OpForm &op, int rank, int nranks, const PlanConfig &cfg_in, Plan *out) {
  if (cfg_in.amin >= 0) return make_plan_with(op, rank, nranks, cfg_in, out);
  // default run length of the window passes: 256 B (amin = 4); at >= 2^29 local amplitudes 1 KB runs (amin = 6)
  // measured 2.5-4 % faster with the wide group (the L2 merges part of the window pass's gathers,
  // profiles/r01_prof_multi16.txt; L=29: 10.2 against 10.6 ms, same time at L=28) -- unless they cost a launch
  PlanConfig c4 = cfg_in;
  c4.amin = 4;
  DNM_TRY(make_plan_with(op, rank, nranks, c4, out));
  if (out->use_tiled && out->n_loc >= 29 && out->cfg.mode == 2 && !cfg_in.swz) {
    PlanConfig c6 = cfg_in;
    c6.amin = 6;
    Plan p6;
    DNM_TRY(make_plan_with(op, rank, nranks, c6, &p6));
    if (p6.local.size() <= out->local.size()) *out = p6;
  }
  return 0;
}

std::string Plan::describe(const OpForm &op) const {
  std::ostringstream os;
  os << "n=" << n << " n_loc=" << n_loc << " rank=" << rank << "/" << nranks
     << " tiled=" << (use_tiled ? 1 : 0) << " B=" << cfg.B << " logR=" << cfg.logR
     << " mode=" << cfg.mode << " masks=" << op.masks.size() << "\n";
  auto dump = [&](const char *kind, const PassSpec &ps, size_t i) {
    os << kind << " pass " << i << ":";
    if (ps.B != cfg.B || (ps.logR && ps.logR != cfg.logR)) os << " B=" << ps.B << " logR=" << (ps.logR ? ps.logR : cfg.logR);
    os << " segs";
    for (int j = 0; j < ps.nseg; ++j)
      os << " [" << ps.seg_pos[j] << "," << ps.seg_pos[j] + ps.seg_len[j] << ")";
    if (ps.glen) os << " xcd-group [" << ps.gpos << "," << ps.gpos + ps.glen << ")";
    os << " diag=" << ps.has_diag << " acc=" << ps.accumulate << " tile_masks=" << ps.tile_masks.size()
       << " gather_masks=" << ps.gather_masks.size();
    if (ps.partner >= 0)
      os << " partner=" << ps.partner << " rows [" << ps.y_off << ",+2^" << ps.n_eff << ") from partner offset "
         << ps.src_off;
    os << "\n";
  };
  for (size_t i = 0; i < local.size(); ++i) dump("local", local[i], i);
  for (size_t i = 0; i < remote.size(); ++i) dump("remote", remote[i], i);
  return os.str();
}

}  // namespace dnm
