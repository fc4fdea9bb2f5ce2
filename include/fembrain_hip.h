/* fembrain_hip.h -- C ABI of libfembrain_hip.so: the MI355X (gfx950) implementation of FemBrain's per-step
 * deformable hot path (element stiffness, sparse assembly, Keff/rhs algebra, Jacobi-PCG) and of the BlobTree
 * field sweep / cell classification / tetrahedral polygonizer.
 *
 * Conventions: plain pointers and sizes, caller-owned host buffers (copied during the call), device memory
 * owned by the handle, one caller thread per handle, no exceptions across the boundary.  Every function
 * returns FB_OK (0) or a negative FB_E* code; fb_last_error() gives the text of the last failure on the
 * calling thread.  Paths cited below are relative to the reference tree's src/ directory.
 */
#ifndef FEMBRAIN_HIP_H
#define FEMBRAIN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define FB_OK 0
#define FB_EINVAL (-1)   /* bad argument (size, index range, unsorted constrained DOFs, ...) */
#define FB_EDEVICE (-2)  /* HIP runtime error / no gfx950 device */
#define FB_ENOMEM (-3)
#define FB_ESOLVER (-4)  /* PCG did not converge (reference: printf + exit(-1), PS_VolumeConservingIntegrator.cpp:203-209) */
#define FB_ECOMM (-5)    /* RCCL not available / communicator error */

/* matrix storage precision (vectors, dot products and all per-element geometry are always fp64) */
#define FB_MATRIX_F32 0 /* north-star fp32 stiffness storage (every product, sum and vector stays fp64) */
#define FB_MATRIX_F64 1 /* reference-width storage, used by the tight parity tests */
/* default (fb_fem_default_params): by size.  fp32 where it buys speed -- from 2 SELL slices per CU on (> 16,384 nodes on 256 CUs: the
 * persistent solver's range, where a third of the matrix stays in LDS; a shard decides by the whole mesh) -- and fp64 below: a system that
 * small is solved from L2 by the two-launch iteration either way, and fp32 storage only costs it accuracy (the 204-DOF disc.1.veg, condition
 * ~1e5: 1e-2 of max|q| after three steps with fp32 values, 2e-5 with fp64).  Decided again at every re-sync.
 * FB_PCG_PERSISTENT asked for explicitly means fp32. */
#define FB_MATRIX_AUTO 2

/* PCG formulation.  Both run the Jacobi-PCG of CGSolver.cpp:129-190 with the exact-residual refresh every 30th
 * iteration.  REFERENCE performs its two reductions per iteration literally (d.q, then sum r^2/diag).  MERGED obtains
 * rho_new = rho - 2 alpha S1 + alpha^2 S2 from sums taken in the SpMV launch, so an iteration is two kernels and one
 * reduction (one RCCL all-reduce when sharded); iterates agree to rounding, checked by the parity tests. */
#define FB_PCG_MERGED 0
#define FB_PCG_REFERENCE 1
/* (value 2, an experimental one-launch-per-iteration form that measured 1.8x slower, was removed in round 3: FB_EINVAL) */
/* PERSISTENT: the whole solve inside ONE launch (fembrain_amd/csrc/pcg_pipe.hip.h): one workgroup per CU, one wavefront per SELL
 * slice, every lane keeps its row's x, r, w, z, s, p, 1/diag in registers.  The iteration is the PIPELINED form of the same
 * Jacobi-PCG (Ghysels & Vanroose 2014: the same iterates in exact arithmetic): its two sums are posted before the product and
 * read after it, so the only wait of an iteration is for the neighbouring workgroups' part of the product's input vector.
 * Every 30th iteration takes the exact residual as CGSolver.cpp:159-166 does.  Unsharded handles (sharded ones: opt-in, see
 * fb_fem_set_sharded_persist), FB_MATRIX_F32 storage, up to
 * 24 slices per CU (~2.2M tets on 256 CUs: one row per lane up to 12 slices per CU, two rows per lane with x, p, z in LDS from 13
 * on); chosen BY DEFAULT from 2 slices per CU on (FB_PCG_MERGED + FEMBRAIN_PCG_PERSIST unset).  Iteration
 * counts equal those of the literal solver within max(3, 2 %) (tests), iterates agree to rounding and are bitwise
 * reproducible however the solve is cut into launches.  If a wait inside the launch times out (FEMBRAIN_PERSIST_TIMEOUT_MS,
 * default 50, 2000 on a sharded handle; the workgroups must all be resident) the solve is repeated with the two-launch iteration and the handle stays
 * with it: fb_step_info.pcg_path = FB_PCG_PATH_FALLBACK, fb_step_info.persist_fallbacks counts (FEMBRAIN_PERSIST_STRICT=1:
 * FB_EDEVICE instead).  ACCURACY LIMIT: the pipelined recurrences stall at a relative residual of ~1e-11 on these systems (the
 * literal ones go on below 1e-12), so solves with a tolerance below 1e-8 (the reference uses 1e-6) run the two-launch solver, and
 * a persistent solve that ends at the iteration cap is checked with ONE exact residual: if the true r . r / diag of its iterate is within a
 * factor of four of what the recurrences carried, the iterate stands (-max_iter iterations, as CGSolver.cpp:189 returns); otherwise the
 * two-launch solver repeats the solve (FB_PCG_PATH_RESOLVED; FEMBRAIN_PERSIST_CAP_CHECK=0: always). */
#define FB_PCG_PERSISTENT 3
/* BLOCK_JACOBI (opt-in; NOT the reference's solver, excluded from parity): PCG with the inverse of every row's 3x3 diagonal block
 * as preconditioner instead of the inverse diagonal.  Same convergence test (on r . B^-1 r).  Unsharded handles.  Where the
 * handle is eligible for the one-row persistent kernel (fp32 matrix, 2..12 slices per CU) the solve runs inside it
 * (k_pcg_pipe<.., BJ>: the pipelined recurrences); otherwise, and below a tolerance of 1e-8, the literal two-launch sequence.
 * Measured on the 1M-tet cube, first step from rest: 1,459 instead of 1,713 iterations at the same time each. */
#define FB_PCG_BLOCK_JACOBI 4

const char* fb_last_error(void);
int fb_device_count(void);
/* name/arch of device `dev` into caller buffers (may be NULL) */
int fb_device_info(int dev, char* name, int name_len, char* arch, int arch_len, int* n_cu);
/* Page-locks caller memory (hipHostRegister) so that the state copies of a step (fb_fem_set_state / fb_fem_get_state /
 * fb_fem_set_external_forces: the q, qvel, qaccel, externalForces arrays IntegratorBase allocates, integratorBase.cpp:36-60) run
 * at PCIe speed instead of through a staging copy.  Optional; unregister before the memory is freed. */
int fb_host_register(void* p, unsigned long long bytes);
int fb_host_unregister(void* p);

/* ------------------------------------------------------------------------------------------------------
 * FEM handle.  One handle = what Deformable::syncForceModel builds (deformable/Deformable.cpp:127-220):
 * TetMesh + CorotationalLinearFEM (3rdparty/vegafem/corotationalLinearFEM/corotationalLinearFEM.cpp:40-146)
 * + consistent mass (volumetricMesh/generateMassMatrix.cpp:33-76) + the integrator state of
 * ImplicitNewmarkSparse / VolumeConservingIntegrator (integrator/implicitNewmarkSparse.cpp:39-83,
 * deformable/PS_VolumeConservingIntegrator.cpp:17-28).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct fb_fem_s* fb_fem_t;

typedef struct fb_fem_params {
  double E, nu, rho;            /* Deformable.cpp:178 uses 1e7, 0.46, 1000 */
  double timestep;              /* Deformable.cpp:113: 0.0333 */
  double damping_mass;          /* Rayleigh c_M, Deformable.cpp:107: 0 */
  double damping_stiffness;     /* Rayleigh c_K, Deformable.cpp:110: 0.01 */
  double cg_eps;                /* PS_VolumeConservingIntegrator.cpp:196: 1e-6 */
  int cg_max_iter;              /* PS_VolumeConservingIntegrator.cpp:197: 10000 */
  int matrix_precision;         /* FB_MATRIX_AUTO (default) / FB_MATRIX_F32 / FB_MATRIX_F64 */
  int device;                   /* HIP device ordinal */
  int pcg_variant;              /* FB_PCG_MERGED (default) / FB_PCG_REFERENCE */
  int spmv_kernel;              /* 0 = choose by size, FB_SPMV_ROWS, FB_SPMV_SPLIT (small meshes: one slice per block) */
  int linear;                   /* non-zero: warp = 0 of CorotationalLinearFEMForceModel (corotationalLinearFEM.cpp:429-453): no rotation
                                 * extraction, K = K0 and f = K0 u.  0 = the corotational model FemBrain uses (warp = 1, its default) */
  int exact_tangent;            /* non-zero: warp = 2 (corotationalLinearFEM.cpp:296-428): the derivative of the element rotation is added
                                 * to the tangent stiffness (f_int is that of warp = 1); costs 144 stored values per element.  0 = FemBrain's */
  int integrator;               /* FB_INTEGRATOR_VOLUME_CONSERVING (0, FemBrain's VolumeConservingIntegrator::DoTimestep) or
                                 * FB_INTEGRATOR_NEWMARK (ImplicitNewmarkSparse::DoTimestep, implicitNewmarkSparse.cpp:183-379) */
  int renumber;                 /* FB_RENUMBER_AUTO (0) / FB_RENUMBER_ON / FB_RENUMBER_OFF: locality renumbering of the nodes inside the handle,
                                 * see below.  FEMBRAIN_RENUMBER=0/1 in the environment overrides */
  int expect_cuts;              /* non-zero: the caller will cut this mesh (CuttableMesh::cut, deformable/CuttableMesh.cpp:283-470: cells erased,
                                 * pieces and nodes appended) and re-sync with fb_fem_resync_delta.  The handle then (1) keeps a quarter of slack
                                 * in every buffer that grows with the mesh (an eighth otherwise) -- or room for reserve_nodes / reserve_elements --
                                 * so that a growing mesh re-allocates every few cuts, not every second one; (2) chooses its internal node order
                                 * at creation under FB_RENUMBER_AUTO (appended nodes make any caller order wide at the first cut: the full
                                 * builder would run then); (3) allocates the re-sync's workspace at creation.  The first fb_fem_resync_delta is
                                 * then as cheap as the fifth.  Unsharded handles. */
  int reserve_nodes;            /* expect_cuts: node / element counts the mesh is expected to reach (0: a quarter more than it has) */
  int reserve_elements;
} fb_fem_params;
/* Node numbering.  FemBrain appends every node a cut creates at the end of the node list (deformable/VolMesh.cpp:1086-1091,
 * 1639-1642) and its shipped tet meshes are TetGen outputs (surface vertices first): numberings in which a node's neighbours lie
 * anywhere in the list.  The handle then works in an INTERNAL slab order of its own (nodes sorted by their rest position quantised
 * to cells, longest axis of the bounding box first; fembrain_amd/csrc/renumber.h) and maps node ids on the way in and out: xyz,
 * tets, constrained DOFs, forces, state and fb_fem_pattern / the block arrays in its order all stay in the CALLER's numbering
 * (columns ascending in the caller's ids); elements keep their order.  Only the rounding of a row's sum in the SpMV changes with
 * the order of its columns.  AUTO renumbers meshes of >= 8,192 nodes whose widest element (largest id difference inside a tet)
 * exceeds min(32,767, nodes / 8) and keeps the caller's order when the slab order is not at least a quarter narrower.  On a sharded
 * handle (fb_fem_create_sharded) the renumbering needs the whole mesh on every rank, and the rank then owns a contiguous range of the
 * INTERNAL order: fb_fem_owned_range reports that range, fb_fem_owned_nodes the caller ids in it, and the state / force arrays are
 * read and written at those nodes' positions.  AUTO on a sharded handle is a collective decision taken before the build (creation
 * and every re-sync): every rank counts the ranks its elements couple it to under the caller's ranges, and the ranks switch to the
 * internal order together when some rank would have more than two neighbour ranks (or, from 8,192 nodes, most of its elements
 * reaching into another rank's range) and every rank was handed the same whole mesh (sizes and a checksum of the element list are
 * compared); ranks that were handed their own elements only keep the caller's numbering.  ON / OFF must be the same on every rank. */
#define FB_RENUMBER_AUTO 0
#define FB_RENUMBER_ON 1
#define FB_RENUMBER_OFF (-1)
#define FB_INTEGRATOR_VOLUME_CONSERVING 0
#define FB_INTEGRATOR_NEWMARK 1
#define FB_SPMV_ROWS 1
#define FB_SPMV_SPLIT 2

/* fills the reference's defaults listed above */
void fb_fem_default_params(fb_fem_params* p);

/* xyz: n_nodes*3 rest positions (fp64); tets: n_tets*4 node ids (0-based);
 * fixed_dofs: ascending, 0-based constrained DOFs (implicitNewmarkSparse.h:78-80), copied. */
int fb_fem_create(fb_fem_t* out, int n_nodes, const double* xyz, int n_tets, const int* tets,
                  int n_fixed_dofs, const int* fixed_dofs, const fb_fem_params* params);

/* Domain-decomposed variant: rank `rank` of `n_ranks` owns the contiguous node range
 * [node_splits[rank], node_splits[rank+1]) of one global mesh (node_splits has n_ranks+1 ascending entries covering
 * [0, n_nodes); NULL = equal split); it assembles every tet touching an owned node and keeps only owned rows, so no
 * force/stiffness reduction is needed.
 * Per-rank ingest: `tets` may be the whole element list (the rank filters it) or ONLY THE RANK'S OWN ELEMENTS -- every tet
 * with at least one owned node, global node ids, in ascending global element order (the order fixes the rounding of the
 * assembled sums).  Only the xyz rows of owned nodes and of nodes those elements reference are read, so a rank may back `xyz` with a sparse
 * mapping.  The plan (pattern, SELL-64, contribution lists) of the owned rows is built on the device.
 * Per PCG iteration: one halo exchange of the search direction and two fp64 scalar all-reduces over `comm`
 * (RCCL); `comm` may be NULL when n_ranks == 1. */
typedef struct fb_comm_s* fb_comm_t;
int fb_fem_create_sharded(fb_fem_t* out, int n_nodes, const double* xyz, int n_tets, const int* tets,
                          int n_fixed_dofs, const int* fixed_dofs, const fb_fem_params* params,
                          int n_ranks, int rank, const int* node_splits, fb_comm_t comm);
int fb_fem_destroy(fb_fem_t h);
/* How a sharded handle runs the two exchanges of a PCG iteration (halo refresh of the search direction, 3-scalar sum):
 *   FB_XCH_COLLECTIVE  RCCL all-reduce + send/recv (or the host-staged test communicator)
 *   FB_XCH_P2P         direct peer-to-peer inboxes over xGMI (HIP IPC), one small kernel per exchange
 *   FB_XCH_P2P_SUMS    as above, but the sums are posted by the last SpMV block and awaited by the vector pass
 *   FB_XCH_P2P_FUSED   both exchanges ride inside the two kernels of the iteration (default when every rank could map
 *                      every peer; FEMBRAIN_P2P=0 keeps the collective library, FEMBRAIN_XCH_MODE=n picks a mode)
 * All modes give bitwise identical iterates (fixed rank-order sums).  fb_fem_transport: 0 for an unsharded handle, else
 * the mode in use.  fb_fem_set_exchange_mode is COLLECTIVE: every rank switches between two solves. */
#define FB_XCH_COLLECTIVE 1
#define FB_XCH_P2P 2
#define FB_XCH_P2P_SUMS 3
#define FB_XCH_P2P_FUSED 4
int fb_fem_transport(fb_fem_t h);
int fb_fem_set_exchange_mode(fb_fem_t h, int mode);
/* The sharded persistent pipelined solver (pcg_shard_box.hip.h): one launch per solve on every rank, halo rows and rank sums crossing
 * the GPUs inside the launches, no exchange kernels or collectives on the path.  OPT-IN (FEMBRAIN_SHARDED_PERSIST=1 when the handle
 * is created; <= 22 slices per CU -- two rows per lane from 12 on -- and <= 16 ranks) and UNMEASURED on multi-GPU hardware.  fb_fem_sharded_persist: 1 when the handle's
 * solves run in it.  fb_fem_set_sharded_persist is COLLECTIVE like fb_fem_set_exchange_mode (0: the two-launch iteration with the
 * exchange mode above); FB_EINVAL when it was never attached, FB_EDEVICE after a launch has timed out (every rank fell back). */
int fb_fem_sharded_persist(fb_fem_t h);
int fb_fem_set_sharded_persist(fb_fem_t h, int on);

/* Rebuild after a topology change (Deformable::syncForceModel after CuttableMesh::cut, main.cpp:614-617):
 * same semantics as destroy + create but keeps the device, parameters and constraints. State is reset. */
int fb_fem_resync(fb_fem_t h, int n_nodes, const double* xyz, int n_tets, const int* tets,
                  int n_fixed_dofs, const int* fixed_dofs);
/* The same on a sharded handle, COLLECTIVE (every rank calls it at the same point, with the arguments fb_fem_create_sharded
 * takes: the whole mesh or the rank's own elements).  node_splits: the new node ranges; NULL keeps the handle's ranges when
 * the node count is unchanged and falls to the equal split otherwise (fb_fem_resync on a sharded handle does exactly that).
 * The exchange mode in use stays; peer-to-peer inboxes are re-attached for the new halo. */
int fb_fem_resync_sharded(fb_fem_t h, int n_nodes, const double* xyz, int n_tets, const int* tets,
                          int n_fixed_dofs, const int* fixed_dofs, const int* node_splits);

/* The same re-sync from a DESCRIPTION of the change (fembrain_amd/csrc/delta.h).  CuttableMesh::cut (CuttableMesh.cpp:283-470) erases
 * the cells its blade crosses keeping the order of the rest (VolMesh.cpp:630, m_vCells.erase), appends their pieces and the new nodes
 * (push_back, VolMesh.cpp:1083-1088) and re-points cells on a split edge in place (:1630-1650); the host passes just that:
 *   removed[n_removed]        ids (in the handle's CURRENT element list) of the elements to erase, ascending
 *   changed_ids[n_changed]    ids of elements that stay in place with new nodes, ascending, none of them removed;
 *   changed_nodes[4 n_changed]  their four node ids
 *   added_tets[4 n_added]     elements appended behind the last one
 *   new_xyz[3 n_new_nodes]    rest positions of nodes appended behind the last one (ids n_nodes, n_nodes + 1, ...)
 *   fixed_dofs                the whole constrained-DOF list of the new mesh, as fb_fem_resync takes it
 * The element list and the rest positions stay on the device; the result is the state fb_fem_resync would leave with the whole new
 * mesh (state reset).  The plan is updated from the plan (block pattern with the pairs of every block, contribution table:
 * fembrain_amd/csrc/delta.hip) instead of built and sorted again: for a handle in the caller's node order the plan is bit for bit the full rebuild's; a renumbered handle
 * (fb_fem_renumbering) puts the new nodes into its slab order under the cell size the order was made with, where a full rebuild
 * would derive a new cell size -- same pattern and values in the caller's ids, roundings of sums aside.  Otherwise (and with
 * FEMBRAIN_RESYNC_DELTA=rebuild) the full builder runs from the device copy of the new mesh.  fb_fem_resync_path: what the last
 * re-sync of the handle did.  Unsharded handles with a device-built plan only (FB_EINVAL otherwise); a bad id is refused before
 * anything changes; after a failure further in the handle is unusable until a full fb_fem_resync succeeds. */
#define FB_RESYNC_FULL 0           /* fb_fem_create / fb_fem_resync: whole mesh from the host */
#define FB_RESYNC_DELTA_MERGED 1   /* fb_fem_resync_delta: the plan updated from the plan */
#define FB_RESYNC_DELTA_REBUILT 2  /* fb_fem_resync_delta: full builder from the device copy of the mesh */
int fb_fem_resync_delta(fb_fem_t h, int n_removed, const int* removed, int n_changed, const int* changed_ids, const int* changed_nodes,
                        int n_added, const int* added_tets, int n_new_nodes, const double* new_xyz, int n_fixed_dofs, const int* fixed_dofs);
int fb_fem_resync_path(fb_fem_t h);


/* Per-element rest-state rebuild on the device (M^-1 rows / volume, corotationalLinearFEM.cpp:66-90 and
 * tetMesh.cpp:184-188) -- the per-step "K0 rebuild" of BASELINE config 4. */
int fb_fem_rebuild_elements(fb_fem_t h);

/* IntegratorBase::SetExternalForces / AddExternalForces / SetExternalForcesToZero (integratorBase.cpp:84-103);
 * f has 3*n_nodes entries (global numbering; a sharded handle reads its owned range). */
int fb_fem_set_external_forces(fb_fem_t h, const double* f);
int fb_fem_add_external_forces(fb_fem_t h, const double* f);
int fb_fem_set_external_forces_zero(fb_fem_t h);
/* constant force on one axis of every node, generated on the device: axis 1, value -10000 is the gravity
 * load of Deformable::timestep (Deformable.cpp:331-338) */
int fb_fem_set_uniform_force(fb_fem_t h, int axis, double value);

typedef struct fb_step_info {
  int cg_iterations;      /* |return value| of CGSolver::SolveLinearSystemWithJacobiPreconditioner (CGSolver.cpp:189) */
  int converged;          /* 1 / 0 */
  double assembly_seconds;/* IntegratorBaseSparse::GetForceAssemblyTime (integratorBaseSparse.h:66), hipEvent-timed */
  double solve_seconds;   /* IntegratorBaseSparse::GetSystemSolveTime (integratorBaseSparse.h:67) */
  double rho0, rho;       /* initial / final sum r^2 / diag */
  int pcg_path;           /* FB_PCG_PATH_*: which solver form ran the (last) solve of this step */
  int persist_fallbacks;  /* solves of this handle so far that a timed-out persistent launch handed to the two-launch form */
  int newton_iterations;  /* FB_INTEGRATOR_NEWMARK: linear solves of this step (implicitNewmarkSparse.cpp:201-379, numIter); 1 otherwise */
} fb_step_info;
#define FB_PCG_PATH_TWO_LAUNCH 0  /* k_spmv + k_cg_fused per iteration (or the literal / block-Jacobi sequences) */
#define FB_PCG_PATH_PERSISTENT 1  /* the persistent launch (FB_PCG_PERSISTENT) */
#define FB_PCG_PATH_FALLBACK 2    /* a persistent launch timed out; the solve was repeated with the two-launch form */
#define FB_PCG_PATH_RESOLVED 3    /* the persistent solve reached the iteration cap; repeated by the two-launch solver, whose result stands */

/* VolumeConservingIntegrator::DoTimestep (PS_VolumeConservingIntegrator.cpp:46-260): assembly, Keff/rhs, PCG,
 * state update.  Returns FB_OK, or FB_ESOLVER when PCG hit cg_max_iter (state is then left unchanged). */
int fb_fem_step(fb_fem_t h, fb_step_info* info);

/* IntegratorBase::GetqState / SetqState / ResetToRest (integratorBase.cpp:105-131); any pointer may be NULL.
 * Arrays are 3*n_nodes long, global numbering; a sharded handle fills / reads only its owned range. */
int fb_fem_get_state(fb_fem_t h, double* q, double* qvel, double* qaccel);
int fb_fem_set_state(fb_fem_t h, const double* q, const double* qvel, const double* qaccel);
int fb_fem_reset(fb_fem_t h);
int fb_fem_set_timestep(fb_fem_t h, double timestep);
int fb_fem_set_damping(fb_fem_t h, double damping_mass, double damping_stiffness);
/* IntegratorBase::SetInternalForceScalingFactor (integratorBase.h; default 1): internal forces and tangent stiffness
 * are multiplied by `factor` from the next assembly on (both are linear in Young's modulus, which is what is scaled) */
int fb_fem_set_internal_force_scaling(fb_fem_t h, double factor);
int fb_fem_set_cg(fb_fem_t h, double eps, int max_iter);
/* Newmark parameters (implicitNewmarkSparse.h: NewmarkBeta 0.25, NewmarkGamma 0.5; IntegratorBase: maxIterations 1, epsilon
 * 1e-6): the Newton loop stops when |residual|^2 / |first residual|^2 < epsilon^2 or after max_newton_iterations.  Each Newton
 * iteration is one assembly + one PCG solve that -- as in the reference -- starts from the previous solution.  The residual is summed
 * over ALL DOFs, the reaction forces at the clamped ones included (implicitNewmarkSparse.cpp:258-262; rounds 2-4 summed the free DOFs
 * only).  Unsharded handles only for more than one iteration (the quotient is a global sum). */
int fb_fem_set_newmark(fb_fem_t h, double beta, double gamma, int max_newton_iterations, double epsilon);
/* IntegratorBaseSparse::setConstrainedDOF (integratorBaseSparse.cpp:73-87) -- takes effect at the next step
 * (the mask is applied when Keff is formed, so unlike the reference no stale systemMatrix can survive) */
int fb_fem_set_constrained_dofs(fb_fem_t h, int n_fixed_dofs, const int* fixed_dofs);

/* Floor-plane collision + velocity rewrite of Deformable::timestep (Deformable.cpp:350-402), on the device:
 * v <- v_t - restitution * v_n for every node (n = +y), q_y clamped so rest_y + q_y >= floor_y.
 * n_collided (may be NULL) counts nodes with rest_y + q_y <= floor_y before the clamp. */
int fb_fem_floor_collision(fb_fem_t h, double floor_y, double restitution, int* n_collided);

/* ---- inspection entry points (what the parity tests compare against the oracle) ---- */
int fb_fem_num_nodes(fb_fem_t h);   /* global */
int fb_fem_num_tets(fb_fem_t h);    /* local (all for an unsharded handle) */
int fb_fem_num_blocks(fb_fem_t h);  /* 3x3 blocks of the stiffness pattern (owned rows) */
int fb_fem_matrix_precision(fb_fem_t h);  /* FB_MATRIX_F32 or FB_MATRIX_F64: the width the matrix values are stored in (what FB_MATRIX_AUTO chose) */
int fb_fem_owned_range(fb_fem_t h, int lo_hi[2]);  /* the node range this handle owns ([0, n_nodes) when unsharded); of the INTERNAL order when renumbered */
/* 1 when the handle works in an internal node order; widest element (largest id difference inside a tet) in the caller's and in the
 * internal order (equal when not renumbered; 0 when never measured: FB_RENUMBER_OFF).  Pointers may be NULL. */
int fb_fem_renumbering(fb_fem_t h, int* span_caller, int* span_internal);
/* a sharded handle's halo: nodes of other ranks its elements touch, and how many ranks own them (0, 0 when unsharded) */
int fb_fem_halo_info(fb_fem_t h, int* n_halo_nodes, int* n_neighbour_ranks);
/* caller ids of the nodes this handle owns, in internal order (hi - lo entries of fb_fem_owned_range; the identity range when not renumbered) */
int fb_fem_owned_nodes(fb_fem_t h, int* ids);
/* node-level pattern, ascending columns per row (corotationalLinearFEM.cpp:163-186, sparseMatrix.cpp:238-262):
 * bptr[n_owned+1], bcol[num_blocks] (global node ids) */
int fb_fem_pattern(fb_fem_t h, int* bptr, int* bcol);
/* per-element K0 = V B^T E B (144 fp64, row-major 12x12) and M^-1 (16 fp64, rows = [grad N_k | N_k(0)]) for
 * elements [first, first+count), computed by the batched MFMA kernel (corotationalLinearFEM.cpp:99-145) */
int fb_fem_element_stiffness(fb_fem_t h, int first, int count, double* K0, double* Minv);
/* raw corotational assembly at displacement u (3*n_nodes): internal force f (3*n_nodes, owned range filled) and
 * K as 9 values per block in fb_fem_pattern order (corotationalLinearFEM.cpp:219-470, warp = 1) */
int fb_fem_assemble(fb_fem_t h, const double* u, double* f, double* K_blocks);
/* Keff (blocks, constrained rows/cols replaced by identity) and right-hand side of the current state, as the
 * next fb_fem_step would solve them (PS_VolumeConservingIntegrator.cpp:84-123,160-162) */
int fb_fem_system(fb_fem_t h, double* Keff_blocks, double* rhs);
/* consistent mass scalar per block (generateMassMatrix.cpp:33-76, tetMesh.cpp:150-182) */
int fb_fem_mass(fb_fem_t h, double* m_blocks);
/* y = Keff x with the device SpMV kernel on the last assembled system (x, y: 3*n_nodes) */
int fb_fem_spmv(fb_fem_t h, const double* x, double* y);
/* Jacobi-PCG (CGSolver.cpp:129-190) on the last assembled system with a caller right-hand side; x starts at 0.
 * iterations_out: + converged / - not converged, as the reference returns. */
int fb_fem_pcg(fb_fem_t h, const double* rhs, double* x, double eps, int max_iter, int* iterations_out);

/* timing helpers for bench.py: `reps` launches of the PCG loop's SpMV kernel (k_spmv<MT,3>: q = A d plus the three
 * merged sums) / `reps` assemblies (k_tet_warp + k_assemble_tets / k_assemble_rows) on the current system; average device seconds per
 * launch measured with HIP events on the handle's stream. */
/* Plan arrays as the device holds them (what the per-step kernels read), for checking the device-side plan builder against
 * the host one (fembrain_hip_testing.h, fb_plan_get -- same names: bptr, bcol, blk_slot, slice_off, colidx, slot_coff,
 * slot_ccnt, contrib).  Copies at most `capacity` int32 words, returns the element count or a negative code. */
long long fb_fem_device_plan_get(fb_fem_t h, const char* name, int* out, long long capacity);
/* 1 if the plan of this handle was built on the device (unsharded handles, unless FEMBRAIN_PLAN_DEVICE=0), else 0 */
int fb_fem_plan_on_device(fb_fem_t h);
/* the assembly kernel of this handle: 2 = element-major, records staged in LDS by one wavefront and mass entries precomputed
 * (k_assemble_tets_st: FB_MATRIX_F32, warp = 1), 1 = element-major with every value wavefront fetching its records
 * (k_assemble_tets; slices of at most 31 slots; also FEMBRAIN_ASM_KERNEL=tets1), 0 = slot-major (k_assemble_rows; also
 * FEMBRAIN_ASM_KERNEL=rows).  All three write the same bits. */
int fb_fem_assembly_kernel(fb_fem_t h);
/* slices of more than 31 slots (hub nodes, hull nodes of a Delaunay mesh): the element-major kernel leaves those to a second launch of the
 * slot-major kernel (same bits); 0 when the slot-major kernel assembles everything anyway */
int fb_fem_assembly_wide_slices(fb_fem_t h);
int fb_fem_time_spmv(fb_fem_t h, int reps, double* seconds_per_spmv);
int fb_fem_time_assembly(fb_fem_t h, int reps, double* seconds_per_assembly);
/* COLLECTIVE on a sharded handle (every rank calls it with the same reps): average device seconds of one halo refresh of
 * a 3-vector and of one 3-scalar global sum, back to back on the handle's stream -- the two exchanges of a PCG
 * iteration, through whichever transport fb_fem_transport() reports.  Both 0 on an unsharded handle. */
int fb_fem_time_exchange(fb_fem_t h, int reps, double* seconds_per_halo, double* seconds_per_sum);
/* average device seconds of forming K0 = V B^T E B for ALL elements with the fp64 MFMA kernel behind
 * fb_fem_element_stiffness (results go to a device scratch, nothing is copied out): what materialising the reference's
 * KElementUndeformed array would cost per rebuild; the per-step path never forms K0. */
int fb_fem_time_element_stiffness(fb_fem_t h, int reps, double* seconds_per_pass);
/* FB_PCG_PERSISTENT (chosen by default where it is faster, see fb_fem_params.pcg_variant): returns 1 if this handle runs its
 * merged iterations inside persistent launches, else 0; wavefronts (= slices) per CU, workgroups, and how many slots of every
 * slice stay resident in LDS for a launch (any pointer may be NULL) */
int fb_fem_persist_info(fb_fem_t h, int* waves_per_cu, int* workgroups, int* lds_slots);
/* What ran: the FB_PCG_PATH_* of the last solve (return value); the persistent kernel instantiation this handle launches
 * ("k_pcg_pipe<float,c16,12,5>", "" when it has none) into name; persistent launches and fallbacks so far; the longest
 * producer list of a workgroup (-1: some workgroup polls all flags).  Any pointer may be NULL. */
int fb_fem_pcg_path(fb_fem_t h, char* name, int name_len, int* persist_launches, int* persist_fallbacks, int* max_producers);
/* A handle whose persistent launch timed out runs the two-launch iteration, but not for ever: after 32 such solves (doubling with
 * every further time-out; FEMBRAIN_PERSIST_REARM=n sets the first wait, 0 = never) an unsharded handle launches the persistent solver
 * again, and every re-sync re-arms it (sharded handles: at a re-sync only, the ranks switch together).  Returns how often this handle
 * was re-armed by the count. */
int fb_fem_persist_rearms(fb_fem_t h);
/* helper tasks of the persistent solver on this mesh (0: none): where a few slices are much wider than the rest (hull nodes of a Delaunay
 * mesh), wavefronts without a slice of their own multiply the upper part of a wide slice's slots and hand the partial sums over in LDS
 * (fembrain_amd/csrc/pcg_pipe.hip.h).  Chosen by the library from the slice widths; FEMBRAIN_PIPE_HELPERS=0/1 overrides. */
int fb_fem_persist_helpers(fb_fem_t h);
/* How the persistent solver lays out the vector its products gather from: 1 = node by node (x, y, z of a node side by side), 0 = three
 * planes.  Decided from the mesh: lines_planes / lines_records (either may be NULL) receive the cache lines the gathers of one matrix slot
 * touch on average in the two forms, sampled on the device when the handle was (re)built -- structured meshes ~12-20 / ~12-15 (planes: each
 * load touches 4 lines), unstructured ones ~80 / ~44 (node by node: 606k-tet Delaunay probe 18.9 -> 16.2 us per PCG iteration).  0 / 0 where
 * it was not measured (handles outside the one-row persistent kernel's range).  FEMBRAIN_PIPE_XYZ=0/1 overrides. */
int fb_fem_persist_gather(fb_fem_t h, double* lines_planes, double* lines_records);
/* average device seconds of ONE persistent launch that starts a solve of the current system and is cut after n_iters
 * iterations (tolerance out of reach), HIP events on the handle's stream around the launch; the difference of two lengths
 * prices an iteration without the launch's fixed cost */
int fb_fem_time_persist(fb_fem_t h, int reps, int n_iters, double* seconds_per_launch);
/* the persistent launches of the SOLVES this handle has made so far (fb_fem_step, fb_fem_pcg; not the timing helper above): how
 * many, their device seconds (HIP events on the handle's stream around each launch) and the PCG iterations they ran */
int fb_fem_persist_stats(fb_fem_t h, int* launches, double* seconds, long long* iterations);
/* algorithmic bytes of ONE Jacobi-PCG iteration on this system: BSR SpMV + the fused lower bound of the vector
 * traffic (9 fp64 vector streams) */
int fb_fem_iteration_bytes(fb_fem_t h, double* bytes);
/* algorithmic bytes moved by one SpMV launch / one assembly on this handle */
int fb_fem_spmv_bytes(fb_fem_t h, double* bytes);
int fb_fem_assembly_bytes(fb_fem_t h, double* bytes);

/* ------------------------------------------------------------------------------------------------------
 * Communicator for sharded handles: RCCL over xGMI, loaded lazily (a 1-GPU process never needs librccl).
 * unique_id: 128 bytes produced by fb_comm_unique_id on rank 0 and broadcast by the launcher.
 * ---------------------------------------------------------------------------------------------------- */
int fb_comm_unique_id(unsigned char id[128]);
int fb_comm_create(fb_comm_t* out, int rank, int n_ranks, const unsigned char id[128], int device);
int fb_comm_destroy(fb_comm_t c);
/* what the communicator is made of: *ranks = its size, *rccl_ranks = the rank count the RCCL communicator itself reports
 * (ncclCommCount; 0 when the communicator has no RCCL handle: the host-staged test transport), *transport = 0 none (one rank, no library),
 * 1 RCCL, 2 host-staged test transport.  Any pointer may be NULL. */
int fb_comm_info(fb_comm_t c, int* ranks, int* rccl_ranks, int* transport);

/* ------------------------------------------------------------------------------------------------------
 * BlobTree field / polygonizer handle = the GPU side of PS::SKETCH::GPUPoly (implicit/OclPolygonizer.h:45-231)
 * and PS::SKETCH::FieldComputer (implicit/FieldComputer.h:31-70).  Input is the LinearBlobTree flat layout
 * (implicit/LinearBlobTree.h:20-82): header 12 floats, 16 per operator, 20 per primitive, 12 per matrix.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct fb_poly_s* fb_poly_t;

/* GPUPoly::setBlob (OclPolygonizer.cpp:1602-1648): deep copy of the four arrays */
int fb_poly_create(fb_poly_t* out, int device, const float* header12, int n_ops, const float* ops16,
                   int n_prims, const float* prims20, int n_mtx, const float* mtx12);
int fb_poly_destroy(fb_poly_t h);

/* Which of the reference's field evaluations the handle follows (the reference has two that disagree):
 *   FB_FIELD_CPU      (default) FieldComputer::fieldValue / computePrimitiveField (implicit/Polygonizer.cpp:1544-2108) without
 *                     its primitive bounding-box cull: range operators sum, binary operators act by their type, instanced
 *                     nodes are evaluated.
 *   FB_FIELD_CPU_BOX  the same WITH computePrimitiveField's isOutsidePrim cull (Polygonizer.cpp:1485-1505,1548-1552), tested
 *                     per point: a primitive contributes 0 outside its box.  prim_boxes6 = lo.xyz, hi.xyz per primitive as
 *                     PrepareAllBoxes makes them (Polygonizer.cpp:210-540, offset ISO_VALUE; BlobReader.h / blobtree.py
 *                     compute them), copied.
 *   FB_FIELD_OPENCL   the OpenCL kernels GPUPoly actually runs (data/opencl/Polygonizer.cl:483-886): ComputeField over the
 *                     traversal route of LinearBlobTree::setTraversalRoute (implicit/LinearBlobTree.cpp:333-429), binary
 *                     operators through ComputeOpField(idxBranchOp, ...) -- the operator's INDEX where its type is meant
 *                     (:825) --, range operators by their own type, instanced nodes 0.  This is the evaluation that wrote the
 *                     reference's shipped outputs (data/models/blobtree/{tumor,peanut,dumbel,dumbelclose,eggshell}.veg): every
 *                     surface vertex of the five files is reproduced in order (tests/golden/surface_*.npz).  FB_EINVAL for
 *                     trees the reference's route builder cannot handle (an operator with two operator children).
 * Takes effect for every later sweep / field / surface call; grids swept before are dropped.  On failure the handle keeps
 * its previous semantics.  The colour pass (fb_poly_read_surface_colors, fb_poly_field_color_array) exists for FB_FIELD_CPU only. */
#define FB_FIELD_CPU 0
#define FB_FIELD_CPU_BOX 1
#define FB_FIELD_OPENCL 2
int fb_poly_set_field_semantics(fb_poly_t h, int semantics, const float* prim_boxes6);
int fb_poly_field_semantics(fb_poly_t h);

/* GPUPoly::computeFieldArray / FieldComputer::field(n,4,xyzf) (OclPolygonizer.cpp:943-986): xyzf is n*4 floats,
 * .w overwritten with the field value */
int fb_poly_field_array(fb_poly_t h, int n, float* xyzf);

/* GPUPoly::computeAllFields (OclPolygonizer.cpp:1358-1426) / FieldComputer::fieldsForVoxelGrid
 * (FieldComputer.cpp:143-231): sweeps the voxel grid of the model's bounding box (header[0..2] .. header[4..6]) at
 * `cellsize` -- points per axis = ceil(extent/cellsize)+2, origin = bbox lower -- and keeps the float4 (x,y,z,f)
 * grid on the device.  dims_out[3] = grid points per axis.  cellsize < 0.01 is refused as GPUPoly::run does. */
int fb_poly_sweep(fb_poly_t h, float cellsize, int dims_out[3]);
/* explicit grid (lower corner, cellsize, point counts) -- used by benches that name the grid size */
int fb_poly_sweep_grid(fb_poly_t h, const float lower[3], float cellsize, const int dims[3]);
/* GPUPoly::readBackVoxelGridSamples (OclPolygonizer.cpp:916-940): xyzf = 4 floats per grid point,
 * index = iz*gx*gy + iy*gx + ix */
int fb_poly_read_grid(fb_poly_t h, float* xyzf);

typedef struct fb_poly_counts {
  int grid[3];
  int n_points, n_cells;
  int n_crossed_edges;    /* = surface vertices, sum of ComputeEdgeTable counts (Polygonizer.cl:1353-1415) */
  int n_surface_cells;    /* cells with 0 < config < 255 */
  int n_included_cells;   /* config != 0 (Tetrahedralizer.cl:3-35) */
  int n_tet_vertices;     /* grid points touched by an included cell */
  int n_tets;             /* 6 per included cell */
  int n_surface_vertices; /* set by fb_poly_surface: = n_crossed_edges (m_ctVertices of GPUPoly::run) */
  int n_surface_indices;  /* set by fb_poly_surface: 3 per triangle (m_ctFaceElements) */
} fb_poly_counts;

/* ComputeEdgeTable + ComputeCellConfigs + TetMeshCells + the exclusive scans, all on the device
 * (OclPolygonizer.cpp:644-757 steps 2,3,5,6 and :762-819) */
int fb_poly_classify(fb_poly_t h, fb_poly_counts* counts);
/* per-point crossing flags (X=4,Y=2,Z=1) / counts and per-cell 8-bit configs, for parity tests; any may be NULL */
int fb_poly_read_classification(fb_poly_t h, unsigned char* edge_flags, unsigned int* edge_counts,
                                unsigned char* cell_configs);
/* GPUPoly::runTetrahedralizer (OclPolygonizer.cpp:762-819; Tetrahedralizer.cl:39-132): compacts the included grid
 * points in grid order and emits 6 tets per included cell; results stay on the device until read back. */
int fb_poly_tetrahedralize(fb_poly_t h, fb_poly_counts* counts);
/* xyz: 3 floats per tet-mesh vertex, tets: 4 uint32 per tet (both sized from fb_poly_counts) */
int fb_poly_read_tetmesh(fb_poly_t h, float* xyz, unsigned int* tets);

/* ---- marching-cubes surface: GPUPoly::run steps 3,4,6,7 (OclPolygonizer.cpp:663-757) ------------------------------
 * Replaces ComputeVertexAttribs (data/opencl/Polygonizer.cl:1429-1561, linear root + forward-difference normal,
 * 4 field evaluations per vertex), ComputeElements (:1610-1670) and the two host scans between them.  Needs
 * fb_poly_classify on a grid swept WITH stored samples.  Vertex colours: fb_poly_read_surface_colors. */
int fb_poly_surface(fb_poly_t h, fb_poly_counts* counts);
/* GPUPoly::readbackMeshV3T3 (OclPolygonizer.cpp:1696-1744): 3 floats per vertex / normal, 3 uint32 per triangle;
 * any pointer may be NULL */
int fb_poly_read_surface(fb_poly_t h, float* xyz, float* normals, unsigned int* indices);
/* the 256 x 16 triangle table (edge ids, 255 = end) and vertex counts the surface pass uses -- the data of
 * src/implicit/_CellConfigTable.h:58-317 / _CellConfigTableCompact.cpp, regenerated from Bloomenthal's cube-table
 * procedure at load time; host only, needs no device */
int fb_poly_cube_table(unsigned char tri[4096], unsigned char nvert[256]);

#define FB_MESH_SURFACE 0
#define FB_MESH_TET 1
/* GPUPoly::applyFemDisplacements (OclPolygonizer.cpp:1543-1596; ApplyVertexDeformations, Polygonizer.cl:1417-1426):
 * out = rest position + (float)displacement for every vertex of the surface or the tet mesh; the rest positions are
 * kept.  n_dof must be 3 x the mesh's vertex count (FB_EINVAL otherwise -- the reference does not check).  The
 * deformed positions stay on the device and are copied to xyz_out when it is not NULL. */
int fb_poly_apply_displacements(fb_poly_t h, int mesh, int n_dof, const double* displacements, float* xyz_out);

/* Render-loop coupling when the FEM mesh is the tet mesh of the SAME grid (fb_poly_tetrahedralize): every surface vertex
 * lies on a grid edge whose two end points are tet-mesh vertices a, b, at the weight t the root finder placed it.
 * out = rest + (float)u_a + t * ((float)u_b - (float)u_a).  (The reference indexes the FEM displacements by the
 * surface-vertex id, OclPolygonizer.cpp:1559-1563, which is only meaningful when both meshes share their vertices;
 * SURVEY 8f-2.)  n_tet_dof = 3 * n_tet_vertices; xyz_out (3 floats per surface vertex) may be NULL. */
int fb_poly_interpolate_displacements(fb_poly_t h, int n_tet_dof, const double* tet_displacements, float* xyz_out);
/* per surface vertex: the pair (a, b) of tet-mesh vertex ids and the weight t (either may be NULL) */
int fb_poly_read_surface_binding(fb_poly_t h, unsigned int* tet_vertex_pairs, float* weights);

/* Vertex colours of the surface mesh (the colour output of ComputeVertexAttribs / FieldComputer::fieldValueAndColor,
 * src/implicit/Polygonizer.cpp:2110-2410): 4 floats (r, g, b, 1) per surface vertex.  Needs fb_poly_surface. */
int fb_poly_read_surface_colors(fb_poly_t h, float* rgba);
/* FieldComputer::fieldValueAndColor for n points: xyzf (x, y, z, f) gets f, rgb 3 floats per point */
int fb_poly_field_color_array(fb_poly_t h, int n, float* xyzf, float* rgb);
/* GPUPoly::computeOffSurfacePointsAndFields (OclPolygonizer.cpp:1045-1107; kernel Polygonizer.cl:1329-1350): for every
 * surface vertex v with normal n the two points v + len n and v - len n with their field values; xyzf_pairs receives
 * 8 floats per vertex (x, y, z, f outside then inside).  Needs fb_poly_surface. */
int fb_poly_off_surface(fb_poly_t h, float len, float* xyzf_pairs);
/* average device seconds of the surface pass (counts + scans + vertex attributes + elements) on the current grid */
int fb_poly_time_surface(fb_poly_t h, int reps, double* seconds);
/* average device seconds of one sweep / one classify+tetrahedralize pipeline on the current grid */
int fb_poly_time_pipeline(fb_poly_t h, int reps, double* sweep_seconds, double* pipeline_seconds);
/* the sweep followed by the float4 (x, y, z, f) grid fb_poly_read_grid returns (16 bytes per point; the sweep itself stores f alone, 4 bytes
 * per point, and the grid is materialised only when it is asked for): SURVEY.md 8d's second figure */
int fb_poly_time_grid(fb_poly_t h, int reps, double* sweep_and_grid_seconds);
/* the same pipeline with HIP events between its stages (the events cost a little: fb_poly_time_pipeline is the rate to quote):
 * average device seconds of [0] the sweep, [1] classification + scans, [2] tet-mesh vertices, [3] tet elements (the dominant kernel:
 * 96 B written per included cell), [4] all of it */
int fb_poly_time_stages(fb_poly_t h, int reps, double seconds[5]);

/* Field grid -> tets -> FEM with no host hop: the tet mesh fb_poly_tetrahedralize left on the device becomes
 * the mesh of a new FEM handle -- node ids copied device to device, float positions widened to double, pattern / SELL-64
 * layout / contribution lists built on the device.  Same result as fb_fem_create on fb_poly_read_tetmesh's arrays.
 * params->device must be the polygonizer's device.  The constrained DOFs come from the host as for fb_fem_create. */
int fb_fem_create_from_poly(fb_fem_t* out, fb_poly_t poly, int n_fixed_dofs, const int* fixed_dofs, const fb_fem_params* params);

/* ---- multi-GPU field path: z-slabs of one grid ---------------------------------------------------------------------
 * The grid's point planes are dealt to the ranks in contiguous runs.  A rank that owns planes [p0, p1) (and the cell layers
 * [p0, min(p1, planes - 1))) sweeps the slab [max(p0 - 1, 0), min(p1 + 1, planes - 1)] -- one plane below and two above,
 * so that the vertex marks of its own planes and of plane p1 are complete -- and runs fb_poly_classify and
 * fb_poly_tetrahedralize on it unchanged.  Point positions are lower + cellsize * GLOBAL index, so field samples, marks
 * and positions are bitwise those of the whole grid.  Vertices are numbered plane by plane: with vertex_base = the number
 * of owned vertices of all lower ranks (one all-gather of fb_poly_slab_counts' first number), fb_poly_read_tetmesh_slab
 * returns the rank's consecutive piece of the whole grid's tet mesh -- the pieces of all ranks concatenated ARE
 * fb_poly_read_tetmesh of the single-GPU run, bit for bit. */
int fb_poly_sweep_slab(fb_poly_t h, const float lower[3], float cellsize, const int dims[3], int z_first, int z_count);
int fb_poly_slab_counts(fb_poly_t h, int own_first_plane, int own_planes, int own_layers, int* n_vertices, int* n_tets);
int fb_poly_read_tetmesh_slab(fb_poly_t h, int own_first_plane, int own_planes, int own_layers, unsigned int vertex_base, float* xyz, unsigned int* tets);

/* ---- cutting tool: scalpel / tet-mesh intersection tests --------------------------------------------------------------------
 * Replaces the device half of PS::FEM::Cutting (src/deformable/Cutting.cpp:87-497) and the four kernels of
 * data/opencl/Cutting.cl (:158-341).  Face f of tet t is face 4 t + f with corners faceMask[f] = {0,1,2} {1,2,3} {2,3,0}
 * {0,1,3}; edge e of tet t is edge 6 t + e with ends {0,1} {1,2} {2,0} {0,3} {1,3} {2,3} (Cutting.cl:174-176, :290-292).
 * fp32 arithmetic, operation for operation that of the reference's IntersectSegmentTriangleF
 * (src/graphics/Intersections.cpp:12-64). */
typedef struct fb_cut_s* fb_cut_t;
#define FB_CUT_FACES 0
#define FB_CUT_EDGES 1
/* Cutting::createMemBuffers (Cutting.cpp:87-167): node positions (3 doubles each, rounded to float as there) and the
 * tets' node ids (4 each).  Ids are checked against n_vertices (FB_EINVAL; the reference does not check). */
int fb_cut_create(fb_cut_t* out, int device, int n_vertices, const double* xyz, int n_tets, const unsigned int* tets);
void fb_cut_destroy(fb_cut_t h);
/* new positions of the same nodes (the deformed mesh) */
int fb_cut_set_vertices(fb_cut_t h, int n_vertices, const double* xyz);
/* Cutting::computeFaceCentroids (Cutting.cpp:253-322; kernel ComputePerTetCentroids): every face flag 1, point = centroid */
int fb_cut_face_centroids(fb_cut_t h);
/* Cutting::computeFaceIntersections (Cutting.cpp:169-251; kernel ComputePerTetFaceIntersections): flag = the scalpel edge
 * s0-s1 (3 doubles each) crosses the face, point = the crossing (w = 1), the centroid otherwise; *n_hits = number of
 * flagged faces (what the reference's sum scan returns).  n_hits may be NULL. */
int fb_cut_face_intersections(fb_cut_t h, const double* s0, const double* s1, int* n_hits);
/* Cutting::computeEdgeIntersections (Cutting.cpp:442-497; kernel ComputePerTetEdgeIntersections): quad12 = the swept quad
 * q0..q3; an edge is tested against triangle (q0, q3, q1), then (q0, q2, q3).  Points of missed edges are (0, 0, 0, 1)
 * (the reference leaves them unwritten). */
int fb_cut_edge_intersections(fb_cut_t h, const double* quad12, int* n_hits);
/* flags (4 or 6 per tet) and points (x, y, z, w per face / edge) of the last pass; either pointer may be NULL */
int fb_cut_read(fb_cut_t h, int what, unsigned int* flags, float* points_xyzw);
/* the flagged faces / edges of the last pass in ascending id order (the compaction the reference's scan prepares):
 * ids and points (4 floats) of at most `capacity` hits, *n_out = number of hits.  ids / points may be NULL to query. */
int fb_cut_read_hits(fb_cut_t h, int what, int capacity, unsigned int* ids, float* points_xyzw, int* n_out);
/* kernel ComputeSegmentTriIntersections (Cutting.cl:321-341; Cutting::computeFaceSegmentIntersectionTest): loose
 * triangles (3 x float4 each) against the segment s0-s1 (3 floats each); out = crossing or (-1, -1, -1, 1) */
int fb_cut_segment_triangles(int device, int n_tris, const float* tri_xyzw, const float* s0, const float* s1, float* points_xyzw);
/* average device milliseconds of one face pass (a = s0, b = s1) or one edge pass (a = quad12, b ignored) */
int fb_cut_time(fb_cut_t h, int what, const double* a, const double* b, int reps, double* ms_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* FEMBRAIN_HIP_H */
