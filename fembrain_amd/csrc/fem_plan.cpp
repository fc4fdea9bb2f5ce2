// Host-side plan builder (see fem_plan.h).  Pure C++, no device calls.
#include "fem_plan.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/fembrain_hip.h"

namespace fb {

std::string& last_error() {
  static thread_local std::string e;
  return e;
}

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

static int owner_of(const std::vector<int>& splits, int g) {
  // splits ascending; owner q has splits[q] <= g < splits[q+1]
  int q = int(std::upper_bound(splits.begin(), splits.end(), g) - splits.begin()) - 1;
  return q;
}

int plan_set_constraints(FemPlan& P, int n_fixed, const int* fixed_dofs) {
  const int r = 3 * P.n_global;
  for (int i = 0; i < n_fixed; i++) {
    if (fixed_dofs[i] < 0 || fixed_dofs[i] >= r) return fail(FB_EINVAL, "constrained DOF %d out of range [0,%d)", fixed_dofs[i], r);
    if (i && fixed_dofs[i] <= fixed_dofs[i - 1]) return fail(FB_EINVAL, "constrained DOFs must be strictly ascending (index %d)", i);
  }
  std::vector<uint8_t> gmask((size_t)r, 1);
  for (int i = 0; i < n_fixed; i++) gmask[fixed_dofs[i]] = 0;
  P.dofmask.assign((size_t)3 * P.n_local, 1);
  P.n_fixed_owned = 0;
  for (int l = 0; l < P.n_local; l++)
    for (int k = 0; k < 3; k++) {
      uint8_t m = gmask[(size_t)3 * P.local2global[l] + k];
      P.dofmask[(size_t)3 * l + k] = m;
      if (!m && l < P.n_owned) P.n_fixed_owned++;
    }
  return FB_OK;
}

int build_fem_plan(FemPlan& P, int n_nodes, int n_tets, const int* tets, int n_fixed, const int* fixed_dofs,
                   int n_ranks, int rank, const int* splits) {
  if (n_nodes <= 0 || n_tets <= 0 || !tets) return fail(FB_EINVAL, "empty mesh (%d nodes, %d tets)", n_nodes, n_tets);
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(FB_EINVAL, "bad rank %d of %d", rank, n_ranks);
  if ((long long)n_tets >= (1LL << 28)) return fail(FB_EINVAL, "too many tets for the packed contribution word");
  P = FemPlan();
  P.n_global = n_nodes; P.n_ranks = n_ranks; P.rank = rank;
  P.splits.assign(n_ranks + 1, 0);
  if (splits) {
    for (int i = 0; i <= n_ranks; i++) P.splits[i] = splits[i];
  } else {
    for (int i = 0; i <= n_ranks; i++) P.splits[i] = int((long long)n_nodes * i / n_ranks);
  }
  if (P.splits[0] != 0 || P.splits[n_ranks] != n_nodes) return fail(FB_EINVAL, "node splits must cover [0,%d)", n_nodes);
  for (int i = 0; i < n_ranks; i++)
    if (P.splits[i + 1] <= P.splits[i]) return fail(FB_EINVAL, "rank %d owns no nodes", i);
  P.node_lo = P.splits[rank]; P.node_hi = P.splits[rank + 1];
  P.n_owned = P.node_hi - P.node_lo;
  for (long long k = 0; k < 4LL * n_tets; k++)
    if (tets[k] < 0 || tets[k] >= n_nodes) return fail(FB_EINVAL, "tet %lld references node %d outside [0,%d)", k / 4, tets[k], n_nodes);

  // local tets: any owned node
  auto owned = [&](int g) { return g >= P.node_lo && g < P.node_hi; };
  for (int e = 0; e < n_tets; e++) {
    const int* t = tets + 4 * (size_t)e;
    if (owned(t[0]) || owned(t[1]) || owned(t[2]) || owned(t[3])) P.tet_global.push_back(e);
  }
  P.n_tets = (int)P.tet_global.size();
  // halo = non-owned nodes of local tets
  std::vector<int> halo;
  for (int le = 0; le < P.n_tets; le++) {
    const int* t = tets + 4 * (size_t)P.tet_global[le];
    for (int i = 0; i < 4; i++)
      if (!owned(t[i])) halo.push_back(t[i]);
  }
  std::sort(halo.begin(), halo.end());
  halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
  P.n_halo = (int)halo.size();
  P.n_local = P.n_owned + P.n_halo;
  P.local2global.resize(P.n_local);
  for (int l = 0; l < P.n_owned; l++) P.local2global[l] = P.node_lo + l;
  for (int h = 0; h < P.n_halo; h++) P.local2global[P.n_owned + h] = halo[h];
  P.halo_off.assign(n_ranks + 1, 0);
  for (int h = 0; h < P.n_halo; h++) P.halo_off[owner_of(P.splits, halo[h]) + 1]++;
  for (int q = 0; q < n_ranks; q++) P.halo_off[q + 1] += P.halo_off[q];
  auto to_local = [&](int g) -> int {
    if (owned(g)) return g - P.node_lo;
    return P.n_owned + int(std::lower_bound(halo.begin(), halo.end(), g) - halo.begin());
  };
  P.tets.resize((size_t)4 * P.n_tets);
  for (int le = 0; le < P.n_tets; le++)
    for (int i = 0; i < 4; i++) P.tets[(size_t)4 * le + i] = to_local(tets[4 * (size_t)P.tet_global[le] + i]);

  // send lists: my owned nodes that share a tet with a node owned by q (that is exactly q's halo inside my range)
  {
    std::vector<std::vector<int>> snd(n_ranks);
    for (int le = 0; le < P.n_tets; le++) {
      const int* t = tets + 4 * (size_t)P.tet_global[le];
      int own[4];
      for (int i = 0; i < 4; i++) own[i] = owned(t[i]) ? rank : owner_of(P.splits, t[i]);
      for (int i = 0; i < 4; i++)
        if (own[i] == rank)
          for (int j = 0; j < 4; j++)
            if (own[j] != rank) snd[own[j]].push_back(t[i] - P.node_lo);
    }
    P.send_off.assign(n_ranks + 1, 0);
    for (int q = 0; q < n_ranks; q++) {
      std::sort(snd[q].begin(), snd[q].end());
      snd[q].erase(std::unique(snd[q].begin(), snd[q].end()), snd[q].end());
      P.send_off[q + 1] = P.send_off[q] + (int)snd[q].size();
      P.send_local.insert(P.send_local.end(), snd[q].begin(), snd[q].end());
    }
  }

  // block pattern of owned rows; columns ascending in GLOBAL id (the reference's order)
  {
    std::vector<int> deg(P.n_owned + 1, 0);
    for (int le = 0; le < P.n_tets; le++)
      for (int i = 0; i < 4; i++) {
        int a = P.tets[(size_t)4 * le + i];
        if (a < P.n_owned) deg[a + 1] += 4;
      }
    for (int a = 0; a < P.n_owned; a++) deg[a + 1] += deg[a];
    std::vector<int> fill(deg.begin(), deg.end() - 1), raw((size_t)deg[P.n_owned]);
    for (int le = 0; le < P.n_tets; le++)
      for (int i = 0; i < 4; i++) {
        int a = P.tets[(size_t)4 * le + i];
        if (a >= P.n_owned) continue;
        for (int j = 0; j < 4; j++) raw[fill[a]++] = P.local2global[P.tets[(size_t)4 * le + j]];
      }
    P.bptr.assign(P.n_owned + 1, 0);
    std::vector<int> gcol;
    gcol.reserve(raw.size() / 4);
    for (int a = 0; a < P.n_owned; a++) {
      auto b = raw.begin() + deg[a], e = raw.begin() + deg[a + 1];
      std::sort(b, e);
      e = std::unique(b, e);
      if (b == e) { gcol.push_back(P.node_lo + a); }  // isolated node: keep a diagonal block so the row is solvable
      else gcol.insert(gcol.end(), b, e);
      P.bptr[a + 1] = (int)gcol.size();
    }
    P.n_blocks = (int)gcol.size();
    P.bcol.resize(P.n_blocks);
    for (int p = 0; p < P.n_blocks; p++) P.bcol[p] = to_local(gcol[p]);
  }

  // SELL-64
  P.n_slices = (P.n_owned + kSliceRows - 1) / kSliceRows;
  P.slice_off.assign(P.n_slices + 1, 0);
  for (int s = 0; s < P.n_slices; s++) {
    int w = 0;
    for (int l = 0; l < kSliceRows; l++) {
      int a = s * kSliceRows + l;
      if (a < P.n_owned) w = std::max(w, P.bptr[a + 1] - P.bptr[a]);
    }
    P.slice_off[s + 1] = P.slice_off[s] + w;
  }
  P.n_slots = P.slice_off[P.n_slices];
  P.colidx.assign((size_t)P.n_slots * kSliceRows, 0);
  P.blk_slot.assign(P.n_blocks, 0);
  for (int s = 0; s < P.n_slices; s++) {
    int w = P.slice_off[s + 1] - P.slice_off[s];
    for (int l = 0; l < kSliceRows; l++) {
      int a = s * kSliceRows + l;
      for (int k = 0; k < w; k++) {
        size_t at = ((size_t)P.slice_off[s] + k) * kSliceRows + l;
        if (a < P.n_owned && k < P.bptr[a + 1] - P.bptr[a]) {
          P.colidx[at] = P.bcol[P.bptr[a] + k];
          P.blk_slot[P.bptr[a] + k] = P.slice_off[s] + k;
        } else {
          P.colidx[at] = a < P.n_owned ? a : 0;  // padding: any valid column, its values stay zero
        }
      }
    }
  }

  // contribution lists
  {
    std::vector<int> cnt(P.n_blocks + 1, 0);
    std::vector<int> elblk((size_t)16 * P.n_tets, -1);
    for (int le = 0; le < P.n_tets; le++)
      for (int i = 0; i < 4; i++) {
        int a = P.tets[(size_t)4 * le + i];
        if (a >= P.n_owned) continue;
        for (int j = 0; j < 4; j++) {
          int gb = P.local2global[P.tets[(size_t)4 * le + j]];
          // binary search by global id inside row a
          int lo = P.bptr[a], hi = P.bptr[a + 1] - 1, pos = -1;
          while (lo <= hi) {
            int m = (lo + hi) >> 1, gm = P.local2global[P.bcol[m]];
            if (gm == gb) { pos = m; break; }
            if (gm < gb) lo = m + 1; else hi = m - 1;
          }
          elblk[(size_t)16 * le + 4 * i + j] = pos;
          cnt[pos + 1]++;
        }
      }
    P.slot_ccnt.assign(P.n_slots, 0);
    for (int a = 0; a < P.n_owned; a++)
      for (int p = P.bptr[a]; p < P.bptr[a + 1]; p++) P.slot_ccnt[P.blk_slot[p]] = std::max(P.slot_ccnt[P.blk_slot[p]], cnt[p + 1]);
    P.slot_coff.assign(P.n_slots, 0);
    long long tot = 0;
    for (int t = 0; t < P.n_slots; t++) { P.slot_coff[t] = (int)tot; tot += P.slot_ccnt[t]; }
    if (tot * kSliceRows >= (1LL << 31)) return fail(FB_EINVAL, "contribution table too large (%lld rows)", tot);
    P.n_crows = (int)tot;
    P.contrib.assign((size_t)P.n_crows * kSliceRows, kNoContrib);
    std::vector<int> used(P.n_blocks, 0);
    for (int le = 0; le < P.n_tets; le++)  // ascending element order inside every block's list
      for (int i = 0; i < 4; i++) {
        int a = P.tets[(size_t)4 * le + i];
        if (a >= P.n_owned) continue;
        for (int j = 0; j < 4; j++) {
          int p = elblk[(size_t)16 * le + 4 * i + j];
          int slot = P.blk_slot[p], lane = a % kSliceRows;
          P.contrib[((size_t)P.slot_coff[slot] + used[p]++) * kSliceRows + lane] = ((uint32_t)le << 4) | (uint32_t)(i << 2) | (uint32_t)j;
        }
      }
  }
  return plan_set_constraints(P, n_fixed, fixed_dofs);
}

}  // namespace fb
