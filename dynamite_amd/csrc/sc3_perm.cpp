// Site relabelling for SpinConserve operators on a bond graph (sc3.h: Sc3Perm): which spins go to the fields
// [T | W | Lo] of the internal layout.  A pair hop whose two spins share the field Lo or W is applied from an LDS tile;
// every other hop is a gathered read of another row or block (sc3g_kernels.hip), so the assignment that leaves the
// fewest, cheapest hops between fields is the fastest.  This is a small graph-partition problem (L <= 64 spins, three
// parts of fixed sizes a, w, L - a - w) solved on the host when the operator is built: simulated annealing over swaps of
// two spins between fields, deterministic (own generator), a few restarts; the identity is kept on ties, so a
// nearest-neighbour chain stays where it is and keeps the chain kernels.
// Nothing in the reference corresponds to this (its kernels gather every column, bpetsc_template_2.c:371-412).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "sc3.h"

namespace dnm {

namespace {

// relative cost of a hop by the fields of its two spins (0 Lo, 1 W, 2 T), from the per-hop times of the passes
// (profiles/r05_kagome_passes.txt): an LDS hop is the unit; a gathered hop is a second read of a tile.
const double HOP_COST[3][3] = {{1.0, 5.0, 5.0}, {5.0, 1.0, 3.5}, {5.0, 3.5, 3.0}};

struct Lcg {
  uint64_t s;
  uint32_t next() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 33);
  }
  double unit() { return next() / 2147483648.0; }
};

}  // namespace

void sc3_choose_perm(int L, int a, int w, int64_t nmasks, const int64_t *masks, bool fix_top, int8_t *site_perm,
                     int32_t *counts) {
  const int t = L - a - w;
  std::vector<int> field_of_bit(L);
  for (int b = 0; b < L; ++b) field_of_bit[b] = b < a ? 0 : (b < a + w ? 1 : 2);
  // the bond graph: pairs with their multiplicity (distinct masks only, so normally 1)
  std::vector<std::vector<int>> adj(L);
  std::vector<std::pair<int, int>> bonds;
  for (int64_t m = 0; m < nmasks; ++m) {
    uint64_t mk = (uint64_t)masks[m];
    // (under XParity a hop that touches spin L-1 comes composed with the global flip: every spin but the pair)
    if (fix_top && L >= 3 && __builtin_popcountll(mk) == L - 2 && !((mk >> (L - 1)) & 1ull))
      mk = ~mk & (((uint64_t)1 << L) - 1);
    if (__builtin_popcountll(mk) != 2) continue;
    const int i = __builtin_ctzll(mk), j = 63 - __builtin_clzll(mk);
    if (j >= L) continue;
    bonds.push_back({i, j});
    adj[i].push_back(j);
    adj[j].push_back(i);
  }
  auto total = [&](const std::vector<int> &f) {
    double c = 0;
    for (auto &b : bonds) c += HOP_COST[f[b.first]][f[b.second]];
    return c;
  };
  std::vector<int> ident(L);
  for (int i = 0; i < L; ++i) ident[i] = field_of_bit[i];
  std::vector<int> best = ident;
  double best_cost = total(ident);
  const double ident_cost = best_cost;
  const int nmov = fix_top ? L - 1 : L;          // spin L-1 keeps its place (the top bit of T) under XParity
  if (!bonds.empty() && t >= 1 && nmov >= 2) {
    Lcg rng{0x9e3779b97f4a7c15ull};
    const int restarts = 6, steps = 40000;
    for (int rs = 0; rs < restarts; ++rs) {
      std::vector<int> f = ident;
      if (rs > 0) {                              // a random start: shuffle the movable spins' fields
        for (int i = nmov - 1; i > 0; --i) std::swap(f[i], f[rng.next() % (uint32_t)(i + 1)]);
      }
      double cur = total(f);
      double temp = 2.0;
      for (int it = 0; it < steps; ++it) {
        const int i = (int)(rng.next() % (uint32_t)nmov), j = (int)(rng.next() % (uint32_t)nmov);
        if (f[i] == f[j]) continue;
        // cost change of swapping the fields of spins i and j: only their own bonds move
        double d = 0;
        for (int n : adj[i]) if (n != j) d += HOP_COST[f[j]][f[n]] - HOP_COST[f[i]][f[n]];
        for (int n : adj[j]) if (n != i) d += HOP_COST[f[i]][f[n]] - HOP_COST[f[j]][f[n]];
        if (d <= 0 || rng.unit() < std::exp(-d / temp)) {
          std::swap(f[i], f[j]);
          cur += d;
          if (cur < best_cost - 1e-9) {
            best_cost = cur;
            best = f;
          }
        }
        temp = std::max(0.05, temp * 0.9998);
      }
    }
  }
  if (!(best_cost < ident_cost - 1e-9)) best = ident;
  // bits inside a field go to its spins in ascending order (spin L-1, if it is in T, gets the top bit)
  int next_bit[3] = {0, a, a + w};
  for (int i = 0; i < L; ++i) site_perm[i] = (int8_t)next_bit[best[i]]++;
  if (counts) {
    for (int q = 0; q < 6; ++q) counts[q] = 0;
    for (auto &b : bonds) {
      const int fi = std::min(best[b.first], best[b.second]), fj = std::max(best[b.first], best[b.second]);
      const int slot = fi == fj ? fi : (fi == 0 ? (fj == 1 ? 3 : 4) : 5);
      ++counts[slot];
    }
  }
}

}  // namespace dnm
