/* fembrain_hip_testing.h -- host-only inspection hooks of libfembrain_hip.so used by the CPU test-suite
 * (no device call is made by any function in this header).  They expose the per-rank plan that
 * fb_fem_create[_sharded] builds -- local numbering, halo/send lists, 3x3-block pattern, SELL-64 layout and the
 * element contribution lists -- so that the partition logic can be checked without a GPU, including with
 * world_size-2 gloo runs.  Not part of the reference-facing surface. */
#ifndef FEMBRAIN_HIP_TESTING_H
#define FEMBRAIN_HIP_TESTING_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fb_plan_s* fb_plan_t;

int fb_plan_create(fb_plan_t* out, int n_nodes, int n_tets, const int* tets, int n_fixed_dofs, const int* fixed_dofs,
                   int n_ranks, int rank, const int* node_splits);
int fb_plan_destroy(fb_plan_t p);
/* info[0..11] = n_owned, n_halo, n_tets, n_blocks, n_slices, n_slots, n_crows, n_send, node_lo, node_hi,
 * n_fixed_owned, n_ranks */
int fb_plan_info(fb_plan_t p, int info[12]);
/* copies the named int32 array into out (capacity in elements); returns the element count or a negative code.
 * names: local2global, halo_off, send_off, send_local, tets, tet_global, bptr, bcol, slice_off, colidx, blk_slot,
 *        slot_coff, slot_ccnt, contrib (uint32 bits), dofmask (one int per DOF) */
int fb_plan_get(fb_plan_t p, const char* name, int* out, size_t capacity);

/* Host restatement of the handle's internal node order (fembrain_amd/csrc/renumber.h: slab order of the rest positions): the caller id of
 * every internal id, and the widest element (largest id difference inside a tet) in the caller's and in that order */
int fb_plan_slab_order(int n_nodes, const double* xyz, int n_tets, const int* tets, int* old_of_new, int* span_caller, int* span_internal);

/* Host restatement of one rank's vote on the node order of a sharded handle under FB_RENUMBER_AUTO (fem.hip vote_shard_order): out[0] =
 * the number of other ranks the elements with a node in this rank's range couple it to under the caller's numbering (-1: bad ranges
 * or node ids), out[1] = elements with a node of this rank, out[2] = those that also have a node of another rank.  node_splits may be
 * NULL (equal ranges). */
int fb_plan_shard_vote(int n_nodes, int n_tets, const int* tets, int n_ranks, int rank, const int* node_splits, int out[3]);

/* Host-staged communicator for tests: processes sharing ONE GPU (or none of them owning more than one) exchange through
 * the POSIX shared-memory segment `shm_name` with a process barrier per collective.  It drives exactly the sharded
 * solver path of fb_fem_create_sharded (halo lists, packing, rank-ordered sums) without RCCL, so the N > 1 path can be
 * run on a one-GPU box.  Every rank passes the same name, n_ranks and outbox_bytes; destroy with fb_comm_destroy. */
struct fb_comm_s;
int fb_comm_create_local(struct fb_comm_s** out, int rank, int n_ranks, const char* shm_name, size_t outbox_bytes, int device);
/* Rank 0 replaces a segment of the same name left by a killed run; every wait is bounded by FEMBRAIN_LOCAL_TIMEOUT_MS
 * (default 20000): a rank whose peers do not come gets FB_ECOMM and poisons the segment, which ends every other rank's
 * wait at once.  fb_comm_test_allgather (host only, no device call): `bytes` (<= 512) from every rank into `all`, in rank
 * order -- the barrier pair every collective of the transport is made of. */
int fb_comm_test_allgather(struct fb_comm_s* c, const void* mine, void* all, size_t bytes);

/* Host-only: compiles a BlobTree (operator walk order, slot allocation, expansion of instanced subtrees) exactly as
 * fb_poly_create does and reports the number of evaluation steps and of per-point value slots; needs no device.
 * Lets the CPU suite check every reference model's tree, including the instanced ones. */
int fb_poly_compile_info(int n_ops, const float* ops16, int n_prims, const float* prims20, int* n_steps, int* n_slots);

#ifdef __cplusplus
}
#endif
#endif
