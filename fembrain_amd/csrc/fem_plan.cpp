// Host-side plan builder (see fem_plan.h).  Pure C++, no device calls.
#include "fem_plan.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

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

// host threads for the row-parallel parts of the plan (FEMBRAIN_PLAN_THREADS overrides; at most 16)
static int plan_threads(int n_rows) {
  int t = (int)std::thread::hardware_concurrency();
  if (const char* e = getenv("FEMBRAIN_PLAN_THREADS")) t = atoi(e);
  t = std::max(1, std::min(t, 16));
  return std::max(1, std::min(t, n_rows / 2048 + 1));
}

template <class F>
static void parallel_for(int T, F f) {
  if (T <= 1) { f(0); return; }
  std::vector<std::thread> th;
  th.reserve(T - 1);
  for (int t = 1; t < T; t++) th.emplace_back(f, t);
  f(0);
  for (auto& x : th) x.join();
}

static int owner_of(const std::vector<int>& splits, int g) {
  // splits ascending; owner q has splits[q] <= g < splits[q+1]
  int q = int(std::upper_bound(splits.begin(), splits.end(), g) - splits.begin()) - 1;
  return q;
}

void shard_neighbour_count(int n_nodes, int n_tets, const int* tets, int n_ranks, int rank, const int* splits, int* neighbours, unsigned long long* list_sum,
                           int* splits_sum, int* own_elements, int* boundary_elements) {
  *neighbours = -1; *list_sum = 0; *splits_sum = 0; *own_elements = 0; *boundary_elements = 0;
  if (!tets || n_tets <= 0 || n_nodes <= 0 || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks) return;
  std::vector<int> sp((size_t)n_ranks + 1);
  for (int i = 0; i <= n_ranks; i++) sp[i] = splits ? splits[i] : int((long long)n_nodes * i / n_ranks);
  if (sp[0] != 0 || sp[n_ranks] != n_nodes) return;
  for (int i = 0; i < n_ranks; i++)
    if (sp[i + 1] <= sp[i]) return;
  const int lo = sp[rank], hi = sp[rank + 1];
  const int T = plan_threads(n_tets / 8);
  std::vector<unsigned long long> masks((size_t)T, 0ULL), sums((size_t)T, 0ULL);
  std::vector<int> bad((size_t)T, 0), n_own((size_t)T, 0), n_bnd((size_t)T, 0);
  parallel_for(T, [&](int t) {
    unsigned long long m = 0, sum = 0;
    int own_e = 0, bnd_e = 0;
    const int e0 = (int)((long long)n_tets * t / T), e1 = (int)((long long)n_tets * (t + 1) / T);
    for (int e = e0; e < e1; e++) {
      const int* v = tets + 4 * (size_t)e;
      sum += (unsigned long long)(unsigned)v[0] * 0x9E3779B1ULL + (unsigned long long)(unsigned)v[1] * 0x85EBCA77ULL + (unsigned long long)(unsigned)v[2] * 0xC2B2AE3DULL +
             (unsigned long long)(unsigned)v[3] * 0x27D4EB2FULL + (unsigned long long)e;
      const bool own = (v[0] >= lo && v[0] < hi) || (v[1] >= lo && v[1] < hi) || (v[2] >= lo && v[2] < hi) || (v[3] >= lo && v[3] < hi);
      if (!own) continue;
      own_e++;
      bool foreign = false;
      for (int k = 0; k < 4; k++) {
        if (v[k] < 0 || v[k] >= n_nodes) { bad[t] = 1; continue; }
        if (v[k] >= lo && v[k] < hi) continue;
        foreign = true;
        m |= 1ULL << owner_of(sp, v[k]);
      }
      bnd_e += foreign ? 1 : 0;
    }
    masks[t] = m; sums[t] = sum; n_own[t] = own_e; n_bnd[t] = bnd_e;
  });
  unsigned long long m = 0;
  bool any_bad = false;
  for (int t = 0; t < T; t++) { m |= masks[t]; *list_sum += sums[t]; any_bad = any_bad || bad[t]; *own_elements += n_own[t]; *boundary_elements += n_bnd[t]; }
  m &= ~(1ULL << rank);
  if (!any_bad) *neighbours = __builtin_popcountll(m);
  for (int i = 0; i <= n_ranks; i++) *splits_sum += sp[i] * (i + 1);
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

int begin_fem_partition(FemPlan& P, int n_nodes, int n_tets, int n_ranks, int rank, const int* splits) {
  if (n_nodes <= 0 || n_tets <= 0) return fail(FB_EINVAL, "empty mesh (%d nodes, %d tets)", n_nodes, n_tets);
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(FB_EINVAL, "bad rank %d of %d", rank, n_ranks);
  if ((long long)n_tets >= (1LL << 28)) return fail(FB_EINVAL, "too many tets for the packed contribution word");
  {
    // a re-sync builds into the same object: the element arrays keep their storage (16 MB of fresh pages at 1M tets cost more
    // than the passes that fill them)
    std::vector<int> keep_tets = std::move(P.tets), keep_ids = std::move(P.tet_global), keep_l2g = std::move(P.local2global);
    P = FemPlan();
    P.tets = std::move(keep_tets); P.tet_global = std::move(keep_ids); P.local2global = std::move(keep_l2g);
    P.tets.clear(); P.tet_global.clear(); P.local2global.clear();
  }
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
  return FB_OK;
}

void set_partition_halo(FemPlan& P, const std::vector<int>& halo) {
  P.n_halo = (int)halo.size();
  P.n_local = P.n_owned + P.n_halo;
  P.local2global.resize(P.n_local);
  for (int l = 0; l < P.n_owned; l++) P.local2global[l] = P.node_lo + l;
  for (int h = 0; h < P.n_halo; h++) P.local2global[P.n_owned + h] = halo[h];
  P.halo_off.assign(P.n_ranks + 1, 0);
  for (int h = 0; h < P.n_halo; h++) P.halo_off[owner_of(P.splits, halo[h]) + 1]++;
  for (int q = 0; q < P.n_ranks; q++) P.halo_off[q + 1] += P.halo_off[q];
}

int build_fem_partition(FemPlan& P, int n_nodes, int n_tets, const int* tets, int n_ranks, int rank, const int* splits, bool need_local_tets) {
  static const bool timing = getenv("FEMBRAIN_TIMING") != nullptr;  // development aid, as in fem.hip's build()
  const auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[fembrain] partition: %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  if (!tets) return fail(FB_EINVAL, "empty mesh (%d nodes, %d tets)", n_nodes, n_tets);
  {
    const int rc = begin_fem_partition(P, n_nodes, n_tets, n_ranks, rank, splits);
    if (rc != FB_OK) return rc;
  }
  auto owned = [&](int g) { return g >= P.node_lo && g < P.node_hi; };
  if (n_ranks == 1) {  // local ids are the global ones
    for (long long k = 0; k < 4LL * n_tets; k++)
      if (tets[k] < 0 || tets[k] >= n_nodes) return fail(FB_EINVAL, "tet %lld references node %d outside [0,%d)", k / 4, tets[k], n_nodes);
    P.tet_global.resize(n_tets);
    for (int e = 0; e < n_tets; e++) P.tet_global[e] = e;
    P.n_tets = n_tets;
    P.n_halo = 0;
    P.n_local = P.n_owned;
    P.local2global.resize(P.n_local);
    for (int l = 0; l < P.n_owned; l++) P.local2global[l] = l;
    P.halo_off.assign(2, 0);
    P.send_off.assign(2, 0);
    P.tets.assign(tets, tets + (size_t)4 * n_tets);
    P.n_owned_corners = 4LL * n_tets;
    return FB_OK;
  }

  lap("reset");
  // Pass 1, element-parallel (contiguous element ranges per host thread, results joined in thread order so nothing depends
  // on the thread count): range check, the elements with an owned node, their foreign nodes (halo candidates) and, per
  // neighbour rank, the owned nodes it will want (send candidates).  A re-sync after a cut runs this on every rank.
  const int T = plan_threads(n_tets / 8);
  auto tet_lo = [&](int t) { return (int)((long long)n_tets * t / T); };
  struct Part {
    std::vector<int> halo;
    std::vector<std::pair<int, int>> send;  // (rank, owned local id)
    long long bad = -1, corners = 0;
    int n_mine = 0;
  };
  std::vector<Part> part(T);
  parallel_for(T, [&](int t) {
    Part& W = part[t];
    for (int e = tet_lo(t); e < tet_lo(t + 1); e++) {
      const int* v = tets + 4 * (size_t)e;
      bool own[4];
      int n_own = 0;
      for (int i = 0; i < 4; i++) {
        if (v[i] < 0 || v[i] >= n_nodes) { if (W.bad < 0) W.bad = 4LL * e + i; own[i] = false; continue; }
        own[i] = owned(v[i]);
        n_own += own[i];
      }
      if (!n_own || W.bad >= 0) continue;
      W.n_mine++;
      W.corners += n_own;
      if (n_own == 4) continue;
      for (int j = 0; j < 4; j++) {
        if (own[j]) continue;
        W.halo.push_back(v[j]);
        const int qj = owner_of(P.splits, v[j]);
        for (int i = 0; i < 4; i++)
          if (own[i]) W.send.emplace_back(qj, v[i] - P.node_lo);
      }
    }
    std::sort(W.halo.begin(), W.halo.end());
    W.halo.erase(std::unique(W.halo.begin(), W.halo.end()), W.halo.end());
    std::sort(W.send.begin(), W.send.end());
    W.send.erase(std::unique(W.send.begin(), W.send.end()), W.send.end());
  });
  lap("pass 1");
  for (int t = 0; t < T; t++)
    if (part[t].bad >= 0) {
      const long long k = part[t].bad;
      return fail(FB_EINVAL, "tet %lld references node %d outside [0,%d)", k / 4, tets[k], n_nodes);
    }
  std::vector<int> halo;
  std::vector<std::pair<int, int>> send;
  P.n_owned_corners = 0;
  std::vector<int> first(T + 1, 0);  // local id of the first element each thread keeps
  for (int t = 0; t < T; t++) {
    first[t + 1] = first[t] + part[t].n_mine;
    halo.insert(halo.end(), part[t].halo.begin(), part[t].halo.end());
    send.insert(send.end(), part[t].send.begin(), part[t].send.end());
    P.n_owned_corners += part[t].corners;
  }
  P.n_tets = first[T];
  // halo = non-owned nodes of local tets, ascending => grouped by owner
  std::sort(halo.begin(), halo.end());
  halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
  set_partition_halo(P, halo);
  // send lists: my owned nodes that share a tet with a node owned by q (that is exactly q's halo inside my range), ascending
  std::sort(send.begin(), send.end());
  send.erase(std::unique(send.begin(), send.end()), send.end());
  P.send_off.assign(n_ranks + 1, 0);
  P.send_local.reserve(send.size());
  for (const auto& qs : send) {
    P.send_off[qs.first + 1]++;
    P.send_local.push_back(qs.second);
  }
  for (int q = 0; q < n_ranks; q++) P.send_off[q + 1] += P.send_off[q];

  lap("halo and send lists");
  // per-rank ingest (the caller passed exactly this rank's elements) feeding the device builder: the elements are numbered
  // locally by a device kernel from the caller's array, nothing to compact
  if (!need_local_tets && P.n_tets == n_tets) return FB_OK;
  // Pass 2, the same element ranges: global ids and local numbering of the kept elements
  P.tet_global.resize((size_t)P.n_tets);
  P.tets.resize((size_t)4 * P.n_tets);
  parallel_for(T, [&](int t) {
    int le = first[t];
    for (int e = tet_lo(t); e < tet_lo(t + 1); e++) {
      const int* v = tets + 4 * (size_t)e;
      if (!(owned(v[0]) || owned(v[1]) || owned(v[2]) || owned(v[3]))) continue;
      P.tet_global[le] = e;
      for (int i = 0; i < 4; i++)
        P.tets[(size_t)4 * le + i] = owned(v[i]) ? v[i] - P.node_lo : P.n_owned + int(std::lower_bound(halo.begin(), halo.end(), v[i]) - halo.begin());
      le++;
    }
  });
  lap("pass 2");
  return FB_OK;
}

int build_fem_plan(FemPlan& P, int n_nodes, int n_tets, const int* tets, int n_fixed, const int* fixed_dofs,
                   int n_ranks, int rank, const int* splits) {
  {
    const int rc = build_fem_partition(P, n_nodes, n_tets, tets, n_ranks, rank, splits, true);
    if (rc != FB_OK) return rc;
  }
  auto owned = [&](int g) { return g >= P.node_lo && g < P.node_hi; };
  const int* halo_b = P.local2global.data() + P.n_owned;
  const int* halo_e = halo_b + P.n_halo;
  auto to_local = [&](int g) -> int {
    if (owned(g)) return g - P.node_lo;
    return P.n_owned + int(std::lower_bound(halo_b, halo_e, g) - halo_b);
  };
  // ---- pattern, SELL layout and contribution lists, row-parallel ------------------------------------------------------
  // Everything below is independent per block row (= owned node), so the rows are dealt to host threads in contiguous
  // ranges; the result does not depend on the number of threads.  (A re-sync after a cut rebuilds the whole plan: at
  // 1M tets this part is what the application waits for.)
  const int T = plan_threads(P.n_owned);
  auto row_lo = [&](int t) { return (int)((long long)P.n_owned * t / T); };

  // incidence lists: for every owned row the (tet << 2 | corner) pairs that touch it, ascending tet
  std::vector<int> inc_ptr(P.n_owned + 1, 0);
  for (int le = 0; le < P.n_tets; le++)
    for (int i = 0; i < 4; i++) {
      const int a = P.tets[(size_t)4 * le + i];
      if (a < P.n_owned) inc_ptr[a + 1]++;
    }
  for (int a = 0; a < P.n_owned; a++) inc_ptr[a + 1] += inc_ptr[a];
  std::vector<uint32_t> inc((size_t)inc_ptr[P.n_owned]);
  {
    std::vector<int> fill(inc_ptr.begin(), inc_ptr.end() - 1);
    for (int le = 0; le < P.n_tets; le++)
      for (int i = 0; i < 4; i++) {
        const int a = P.tets[(size_t)4 * le + i];
        if (a < P.n_owned) inc[fill[a]++] = ((uint32_t)le << 2) | (uint32_t)i;
      }
  }

  // block pattern of owned rows; columns ascending in GLOBAL id (the reference's order)
  std::vector<int> gcol;  // global column id of every block
  {
    std::vector<std::vector<int>> part(T);
    std::vector<int> ucnt(P.n_owned, 0);
    parallel_for(T, [&](int t) {
      std::vector<int> tmp;
      std::vector<int>& out = part[t];
      for (int a = row_lo(t); a < row_lo(t + 1); a++) {
        tmp.clear();
        for (int q = inc_ptr[a]; q < inc_ptr[a + 1]; q++) {
          const int le = (int)(inc[q] >> 2);
          for (int j = 0; j < 4; j++) tmp.push_back(P.local2global[P.tets[(size_t)4 * le + j]]);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        if (tmp.empty()) tmp.push_back(P.node_lo + a);  // isolated node: keep a diagonal block so the row is solvable
        ucnt[a] = (int)tmp.size();
        out.insert(out.end(), tmp.begin(), tmp.end());
      }
    });
    P.bptr.assign(P.n_owned + 1, 0);
    for (int a = 0; a < P.n_owned; a++) P.bptr[a + 1] = P.bptr[a] + ucnt[a];
    P.n_blocks = P.bptr[P.n_owned];
    gcol.resize(P.n_blocks);
    P.bcol.resize(P.n_blocks);
    parallel_for(T, [&](int t) {
      std::copy(part[t].begin(), part[t].end(), gcol.begin() + P.bptr[row_lo(t)]);
      for (int p = P.bptr[row_lo(t)]; p < P.bptr[row_lo(t + 1)]; p++) P.bcol[p] = to_local(gcol[p]);
    });
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
  const int TS = std::min(T, std::max(1, P.n_slices));
  auto slice_lo = [&](int t) { return (int)((long long)P.n_slices * t / TS); };
  parallel_for(TS, [&](int t) {
    for (int s = slice_lo(t); s < slice_lo(t + 1); s++) {
      const int w = P.slice_off[s + 1] - P.slice_off[s];
      for (int l = 0; l < kSliceRows; l++) {
        const int a = s * kSliceRows + l;
        for (int k = 0; k < w; k++) {
          const size_t at = ((size_t)P.slice_off[s] + k) * kSliceRows + l;
          if (a < P.n_owned && k < P.bptr[a + 1] - P.bptr[a]) {
            P.colidx[at] = P.bcol[P.bptr[a] + k];
            P.blk_slot[P.bptr[a] + k] = P.slice_off[s] + k;
          } else {
            P.colidx[at] = a < P.n_owned ? a : 0;  // padding: any valid column, its values stay zero
          }
        }
      }
    }
  });

  // contribution lists
  {
    std::vector<int> cnt(P.n_blocks, 0);
    std::vector<uint16_t> where((size_t)4 * inc.size());  // block of (incidence entry, j) as its offset inside the row
    parallel_for(T, [&](int t) {
      for (int a = row_lo(t); a < row_lo(t + 1); a++) {
        const int* cb = gcol.data() + P.bptr[a];
        const int* ce = gcol.data() + P.bptr[a + 1];
        for (int q = inc_ptr[a]; q < inc_ptr[a + 1]; q++) {
          const int le = (int)(inc[q] >> 2);
          for (int j = 0; j < 4; j++) {
            const int gb = P.local2global[P.tets[(size_t)4 * le + j]];
            const int k = (int)(std::lower_bound(cb, ce, gb) - cb);
            where[(size_t)4 * q + j] = (uint16_t)k;
            cnt[P.bptr[a] + k]++;
          }
        }
      }
    });
    for (int a = 0; a < P.n_owned; a++)
      if (P.bptr[a + 1] - P.bptr[a] > 65535) return fail(FB_EINVAL, "node %d has more than 65535 neighbours", P.node_lo + a);
    P.slot_ccnt.assign(P.n_slots, 0);
    parallel_for(TS, [&](int t) {
      for (int s = slice_lo(t); s < slice_lo(t + 1); s++)
        for (int l = 0; l < kSliceRows; l++) {
          const int a = s * kSliceRows + l;
          if (a >= P.n_owned) break;
          for (int p = P.bptr[a]; p < P.bptr[a + 1]; p++) P.slot_ccnt[P.blk_slot[p]] = std::max(P.slot_ccnt[P.blk_slot[p]], cnt[p]);
        }
    });
    P.slot_coff.assign(P.n_slots, 0);
    long long tot = 0;
    for (int t = 0; t < P.n_slots; t++) { P.slot_coff[t] = (int)tot; tot += P.slot_ccnt[t]; }
    if (tot * kSliceRows >= (1LL << 31)) return fail(FB_EINVAL, "contribution table too large (%lld rows)", tot);
    P.n_crows = (int)tot;
    P.contrib.resize((size_t)P.n_crows * kSliceRows);
    parallel_for(T, [&](int t) {  // pad first, every thread its share
      const size_t n = P.contrib.size(), lo = n * t / T, hi = n * (t + 1) / T;
      std::fill(P.contrib.begin() + lo, P.contrib.begin() + hi, kNoContrib);
    });
    parallel_for(T, [&](int t) {  // ascending element order inside every block's list (inc is ascending in the tet)
      std::vector<int> used;
      for (int a = row_lo(t); a < row_lo(t + 1); a++) {
        used.assign(P.bptr[a + 1] - P.bptr[a], 0);
        const int lane = a % kSliceRows;
        for (int q = inc_ptr[a]; q < inc_ptr[a + 1]; q++) {
          const uint32_t le = inc[q] >> 2, i = inc[q] & 3u;
          for (uint32_t j = 0; j < 4; j++) {
            const int k = where[(size_t)4 * q + j];
            const int slot = P.blk_slot[P.bptr[a] + k];
            P.contrib[((size_t)P.slot_coff[slot] + used[k]++) * kSliceRows + lane] = (le << 4) | (i << 2) | j;
          }
        }
      }
    });
  }
  return plan_set_constraints(P, n_fixed, fixed_dofs);
}

}  // namespace fb
