// Host-only test hooks over FemPlan (include/fembrain_hip_testing.h).
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fembrain_hip.h"
#include "../../include/fembrain_hip_testing.h"
#include "fem_plan.h"
#include "renumber.h"

namespace fb {
int fail(int code, const char* fmt, ...);
}

struct fb_plan_s {
  fb::FemPlan plan;
};

extern "C" {

int fb_plan_create(fb_plan_t* out, int n_nodes, int n_tets, const int* tets, int n_fixed_dofs, const int* fixed_dofs, int n_ranks,
                   int rank, const int* node_splits) {
  if (!out) return fb::fail(FB_EINVAL, "null output");
  fb_plan_s* p = new fb_plan_s;
  int rc = fb::build_fem_plan(p->plan, n_nodes, n_tets, tets, n_fixed_dofs, fixed_dofs, n_ranks, rank, node_splits);
  if (rc != FB_OK) {
    delete p;
    return rc;
  }
  *out = p;
  return FB_OK;
}

int fb_plan_destroy(fb_plan_t p) {
  delete p;
  return FB_OK;
}

int fb_plan_info(fb_plan_t p, int info[12]) {
  if (!p || !info) return fb::fail(FB_EINVAL, "null argument");
  const fb::FemPlan& P = p->plan;
  const int v[12] = {P.n_owned, P.n_halo, P.n_tets, P.n_blocks, P.n_slices, P.n_slots, P.n_crows, (int)P.send_local.size(),
                     P.node_lo, P.node_hi, P.n_fixed_owned, P.n_ranks};
  memcpy(info, v, sizeof v);
  return FB_OK;
}

int fb_plan_get(fb_plan_t p, const char* name, int* out, size_t capacity) {
  if (!p || !name) return fb::fail(FB_EINVAL, "null argument");
  const fb::FemPlan& P = p->plan;
  const std::string n(name);
  const std::vector<int>* v = nullptr;
  std::vector<int> tmp;
  if (n == "local2global") v = &P.local2global;
  else if (n == "halo_off") v = &P.halo_off;
  else if (n == "send_off") v = &P.send_off;
  else if (n == "send_local") v = &P.send_local;
  else if (n == "tets") v = &P.tets;
  else if (n == "tet_global") v = &P.tet_global;
  else if (n == "bptr") v = &P.bptr;
  else if (n == "bcol") v = &P.bcol;
  else if (n == "slice_off") v = &P.slice_off;
  else if (n == "colidx") v = &P.colidx;
  else if (n == "blk_slot") v = &P.blk_slot;
  else if (n == "slot_coff") v = &P.slot_coff;
  else if (n == "slot_ccnt") v = &P.slot_ccnt;
  else if (n == "contrib") { tmp.assign(P.contrib.begin(), P.contrib.end()); v = &tmp; }
  else if (n == "dofmask") { tmp.assign(P.dofmask.begin(), P.dofmask.end()); v = &tmp; }
  else return fb::fail(FB_EINVAL, "unknown plan array '%s'", name);
  if (out) {
    if (capacity < v->size()) return fb::fail(FB_EINVAL, "capacity %zu < %zu", capacity, v->size());
    memcpy(out, v->data(), v->size() * sizeof(int));
  }
  return (int)v->size();
}

int fb_plan_slab_order(int n_nodes, const double* xyz, int n_tets, const int* tets, int* old_of_new, int* span_caller, int* span_internal) {
  if (!xyz || !tets || !old_of_new || n_nodes < 1 || n_tets < 0) return fb::fail(FB_EINVAL, "bad argument");
  std::vector<int> o;
  const int rc = fb::host_slab_order(n_nodes, xyz, n_tets, tets, o, span_caller, span_internal);
  if (rc != FB_OK) return rc;
  memcpy(old_of_new, o.data(), sizeof(int) * (size_t)n_nodes);
  return FB_OK;
}

int fb_plan_shard_vote(int n_nodes, int n_tets, const int* tets, int n_ranks, int rank, const int* node_splits, int out[3]) {
  if (!tets || !out) return fb::fail(FB_EINVAL, "null argument");
  unsigned long long sum = 0;
  int splits_sum = 0;
  fb::shard_neighbour_count(n_nodes, n_tets, tets, n_ranks, rank, node_splits, &out[0], &sum, &splits_sum, &out[1], &out[2]);
  return FB_OK;
}

}  // extern "C"
