// Adaptors that DERIVE FROM THE REFERENCE'S OWN ABSTRACT CLASSES, so that FemBrain's host code (Deformable.cpp and anything
// else written against VegaFEM's ForceModel / IntegratorBaseSparse pointers) links against the MI355X path unchanged.
//
// Include this header from inside the reference tree: it needs the reference's own headers on the include path
//   forceModel.h            src/3rdparty/vegafem/forceModel/forceModel.h:42-67          (class ForceModel)
//   integratorBaseSparse.h  src/3rdparty/vegafem/integrator/integratorBaseSparse.h:45-84 (class IntegratorBaseSparse)
//   sparseMatrix.h          src/3rdparty/vegafem/sparseMatrix/sparseMatrix.h:108-381
// and links with -lfembrain_hip.  Written in C++98 (the reference is built with -std=gnu++98 / c++0x).
//
//   PS::FEM::HipCorotationalForceModel    : public ForceModel
//       replaces CorotationalLinearFEM + CorotationalLinearFEMForceModel (elasticForceModel/corotationalLinearFEMForceModel.h):
//       GetInternalForce / GetTangentStiffnessMatrixTopology / GetTangentStiffnessMatrix / GetForceAndMatrix fill the caller's
//       reference SparseMatrix -- same pattern, same ascending column order (corotationalLinearFEM.cpp:163-186).
//       Two device handles: the one the integrator steps on stores the matrix as fp32 (FB_MATRIX_F32: the measured path --
//       persistent pipelined PCG where the mesh fits it); a matrix handed to HOST code comes from a second handle with
//       reference-width storage (FB_MATRIX_F64) that is created the first time a matrix is asked for, so a host that only
//       steps never pays for it
//   PS::FEM::HipVolumeConservingIntegrator : public IntegratorBaseSparse
//       replaces VolumeConservingIntegrator (src/deformable/PS_VolumeConservingIntegrator.h:17-37) the way the reference's own
//       OclVolConservedIntegrator (src/deformable/OclVolConservedIntegrator.h:41-69) was meant to: same constructor
//       arguments, DoTimestep() runs assembly + Jacobi-PCG + state update on the device; q / qvel / qaccel / externalForces
//       stay the base class's host arrays (Getq(), SetqState() ... are non-virtual and keep working on them), copied across the
//       boundary once per step -- the arrays are page-locked (fb_host_register) so the five 3n-vectors move at PCIe speed.
//       maxIterations / epsilon are the Newton-loop arguments of ImplicitNewmarkSparse (PS_VolumeConservingIntegrator.cpp:156,240);
//       FemBrain passes 1 / 1e-6 (Deformable.cpp:205-214) and one pass is what is built; the PCG tolerance is the reference's
//       hard-coded 1e-6 / 10000 (PS_VolumeConservingIntegrator.cpp:196-197)
//   PS::FEM::hipBlackBoxProduct            : CGSolver::blackBoxProductType (sparseSolver/CGSolver.h:65-66) over fb_fem_spmv
#ifndef FEMBRAIN_VEGA_ADAPTORS_H
#define FEMBRAIN_VEGA_ADAPTORS_H

#include <stdio.h>
#include <string.h>

#include <vector>

#include "forceModel.h"
#include "integratorBaseSparse.h"
#include "sparseMatrix.h"

#include "../fembrain_hip.h"

namespace PS {
namespace FEM {

class HipCorotationalForceModel : public ForceModel {
 public:
  // mesh: anything with the VolumetricMesh accessors (volumetricMesh.h:95-100), e.g. the reference's TetMesh
  template <class Mesh>
  HipCorotationalForceModel(const Mesh* mesh, double E, double nu, double rho, int warp = 1, int device = 0) : h_(NULL), h64_(NULL) {
    const int nv = mesh->getNumVertices(), ne = mesh->getNumElements();
    std::vector<double> xyz(3 * (size_t)nv);
    std::vector<int> tets(4 * (size_t)ne);
    for (int i = 0; i < nv; i++)
      for (int k = 0; k < 3; k++) xyz[3 * (size_t)i + k] = (*mesh->getVertex(i))[k];
    for (int e = 0; e < ne; e++)
      for (int k = 0; k < 4; k++) tets[4 * (size_t)e + k] = mesh->getVertexIndex(e, k);
    xyz_.swap(xyz); tets_.swap(tets);
    create(nv, ne, E, nu, rho, warp, device);
  }
  HipCorotationalForceModel(int nv, const double* xyz, int ne, const int* tets, double E, double nu, double rho, int warp = 1, int device = 0)
      : h_(NULL), h64_(NULL), xyz_(xyz, xyz + 3 * (size_t)nv), tets_(tets, tets + 4 * (size_t)ne) {
    create(nv, ne, E, nu, rho, warp, device);
  }
  virtual ~HipCorotationalForceModel() {
    fb_fem_destroy(h_);
    fb_fem_destroy(h64_);
  }
  bool ok() const { return h_ != NULL; }
  // the handle the integrator steps on (fp32-stored matrix)
  fb_fem_t handle() const { return h_; }
  // the handle matrices for host code come from (fp64-stored); NULL until one was asked for
  fb_fem_t matrixHandle() const { return h64_; }

  // f_int is formed in fp64 from the fp64 geometry whatever the matrix storage is: the stepping handle serves it
  virtual void GetInternalForce(double* u, double* internalForces) { check(fb_fem_assemble(h_, u, internalForces, NULL), "GetInternalForce"); }

  // callee allocates, caller deletes (forceModel.h:52)
  virtual void GetTangentStiffnessMatrixTopology(SparseMatrix** tangentStiffnessMatrix) {
    SparseMatrixOutline outline(r);
    double zero[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < r / 3; a++)
      for (int p = bptr_[a]; p < bptr_[a + 1]; p++) outline.AddBlock3x3Entry(a, bcol_[p], zero);  // block indices (sparseMatrix.cpp:158-163)
    *tangentStiffnessMatrix = new SparseMatrix(&outline);
  }

  virtual void GetTangentStiffnessMatrix(double* u, SparseMatrix* K) { GetForceAndMatrix(u, NULL, K); }

  virtual void GetForceAndMatrix(double* u, double* internalForces, SparseMatrix* K) {
    if (!K) { GetInternalForce(u, internalForces); return; }
    if (!h64_ && !create64()) return;
    blocks_.resize(9 * bcol_.size());
    check(fb_fem_assemble(h64_, u, internalForces, &blocks_[0]), "GetForceAndMatrix");
    // the reference matrix keeps each row's columns ascending (sparseMatrix.cpp:238-262) = the order of fb_fem_pattern
    for (int a = 0; a < r / 3; a++)
      for (int p = bptr_[a]; p < bptr_[a + 1]; p++)
        for (int k = 0; k < 3; k++)
          for (int l = 0; l < 3; l++) K->SetEntry(3 * a + k, 3 * (p - bptr_[a]) + l, blocks_[9 * (size_t)p + 3 * k + l]);
  }

 private:
  void create(int nv, int ne, double E, double nu, double rho, int warp, int device) {
    r = 3 * nv;
    nv_ = nv; ne_ = ne;
    fb_fem_default_params(&prm_);
    prm_.E = E; prm_.nu = nu; prm_.rho = rho; prm_.device = device;
    prm_.matrix_precision = FB_MATRIX_AUTO;  // the handle that steps (fp32 values from 2 slices per CU on); host code gets its matrices from create64()
    prm_.linear = warp == 0 ? 1 : 0;
    prm_.exact_tangent = warp == 2 ? 1 : 0;
    if (fb_fem_create(&h_, nv, &xyz_[0], ne, &tets_[0], 0, NULL, &prm_) != FB_OK) {
      fprintf(stderr, "HipCorotationalForceModel: %s\n", fb_last_error());
      h_ = NULL;
      return;
    }
    bptr_.resize((size_t)nv + 1);
    bcol_.resize((size_t)fb_fem_num_blocks(h_));
    check(fb_fem_pattern(h_, &bptr_[0], &bcol_[0]), "pattern");
  }
  bool create64() {
    fb_fem_params p = prm_;
    p.matrix_precision = FB_MATRIX_F64;  // a force model hands its matrix to host code: reference-width values
    if (fb_fem_create(&h64_, nv_, &xyz_[0], ne_, &tets_[0], 0, NULL, &p) != FB_OK) {
      fprintf(stderr, "HipCorotationalForceModel (matrix handle): %s\n", fb_last_error());
      h64_ = NULL;
      return false;
    }
    return true;
  }
  static void check(int rc, const char* what) {
    if (rc != FB_OK) fprintf(stderr, "HipCorotationalForceModel::%s: %s\n", what, fb_last_error());
  }
  fb_fem_t h_, h64_;
  fb_fem_params prm_;
  int nv_, ne_;
  std::vector<double> xyz_;
  std::vector<int> tets_;
  std::vector<int> bptr_, bcol_;
  std::vector<double> blocks_;
};

// CGSolver(int n, blackBoxProductType, void* data): data = the fb_fem_t whose assembled Keff multiplies
inline void hipBlackBoxProduct(const void* data, const double* x, double* Ax) {
  if (fb_fem_spmv((fb_fem_t)data, x, Ax) != FB_OK) fprintf(stderr, "hipBlackBoxProduct: %s\n", fb_last_error());
}

class HipVolumeConservingIntegrator : public IntegratorBaseSparse {
 public:
  // the argument list of VolumeConservingIntegrator / ImplicitNewmarkSparse (implicitNewmarkSparse.h:78-86).  The element data
  // come from the HIP force model (same mesh); massMatrix is used by the base class for GetTotalMass / GetKineticEnergy.
  HipVolumeConservingIntegrator(int r_, double timestep_, SparseMatrix* massMatrix_, HipCorotationalForceModel* forceModel_, int /*positiveDefiniteSolver*/ = 0,
                                int numConstrainedDOFs_ = 0, int* constrainedDOFs_ = NULL, double dampingMassCoef_ = 0.0, double dampingStiffnessCoef_ = 0.0,
                                int /*maxIterations*/ = 1, double /*epsilon*/ = 1E-6, int /*numSolverThreads*/ = 0)
      : IntegratorBaseSparse(r_, timestep_, massMatrix_, forceModel_, numConstrainedDOFs_, constrainedDOFs_, dampingMassCoef_, dampingStiffnessCoef_),
        h_(forceModel_->handle()), iterations_(0), pcgPath_(0), ok_(forceModel_->handle() != NULL) {
    ok_ = ok_ && fb_fem_set_timestep(h_, timestep_) == FB_OK;
    ok_ = ok_ && fb_fem_set_damping(h_, dampingMassCoef_, dampingStiffnessCoef_) == FB_OK;
    ok_ = ok_ && fb_fem_set_constrained_dofs(h_, numConstrainedDOFs_, constrainedDOFs_) == FB_OK;
    if (!ok_) fprintf(stderr, "HipVolumeConservingIntegrator: %s\n", fb_last_error());
    // the base class's host arrays cross the boundary every step: page-locked, they move at PCIe speed (not fatal if refused)
    double* arrs[4] = {q, qvel, qaccel, externalForces};
    for (int i = 0; i < 4; i++) pinned_[i] = fb_host_register(arrs[i], sizeof(double) * (size_t)r_) == FB_OK ? arrs[i] : NULL;
  }
  virtual ~HipVolumeConservingIntegrator() {
    for (int i = 0; i < 4; i++)
      if (pinned_[i]) fb_host_unregister(pinned_[i]);
  }
  // every setter the constructor made was accepted (a device, valid DOF list ...)
  bool ok() const { return ok_; }

  // sets q and (optionally) qvel, as OclVolConservedIntegrator / ImplicitNewmarkSparse::SetState do for the step that follows
  virtual int SetState(double* q_, double* qvel_ = NULL) {
    memcpy(q, q_, sizeof(double) * r);
    if (qvel_) memcpy(qvel, qvel_, sizeof(double) * r);
    else memset(qvel, 0, sizeof(double) * r);
    for (int i = 0; i < numConstrainedDOFs; i++) q[constrainedDOFs[i]] = qvel[constrainedDOFs[i]] = 0.0;
    return 0;
  }

  virtual bool setConstrainedDOF(int num, int* arr) {
    const bool ok = IntegratorBaseSparse::setConstrainedDOF(num, arr);
    return ok && fb_fem_set_constrained_dofs(h_, num, arr) == FB_OK;
  }
  virtual void SetTimestep(double t) { timestep = t; fb_fem_set_timestep(h_, t); }
  virtual void SetInternalForceScalingFactor(double f) { internalForceScalingFactor = f; fb_fem_set_internal_force_scaling(h_, f); }

  // 0 ok, 1 solver failure (integratorBaseSparse.h:62-64)
  virtual int DoTimestep() {
    fb_fem_set_damping(h_, dampingMassCoef, dampingStiffnessCoef);  // the base class setters are inline and not virtual
    if (fb_fem_set_state(h_, q, qvel, qaccel) != FB_OK || fb_fem_set_external_forces(h_, externalForces) != FB_OK) return 1;
    fb_step_info info;
    memset(&info, 0, sizeof info);  // (a step refused before it starts fills nothing in)
    const int rc = fb_fem_step(h_, &info);
    forceAssemblyTime = info.assembly_seconds;
    systemSolveTime = info.solve_seconds;
    iterations_ = info.cg_iterations;
    pcgPath_ = info.pcg_path;
    if (rc != FB_OK) {
      printf("Error: PCG sparse solver returned non-zero exit status %d.\n", -info.cg_iterations);  // the reference's message; it then exit(-1)s
      return 1;
    }
    return fb_fem_get_state(h_, q, qvel, qaccel) == FB_OK ? 0 : 1;
  }
  int lastIterations() const { return iterations_; }
  int lastPcgPath() const { return pcgPath_; }  // FB_PCG_PATH_*

 private:
  fb_fem_t h_;
  int iterations_, pcgPath_;
  bool ok_;
  double* pinned_[4];
};

}  // namespace FEM
}  // namespace PS
#endif
