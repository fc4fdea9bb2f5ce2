// C++ adaptors with the reference's class surfaces over the C ABI (fembrain_hip.h).  Header-only; link with
// -lfembrain_hip.  Scene-graph / GL members of the originals (SGMesh base, draw(), pickVertex, Bullet shapes) stay in
// the host application: these classes carry the simulation state and the per-step hot path only.
//
//   PS::FEM::HipIntegrator  <-> VolumeConservingIntegrator / ImplicitNewmarkSparse / IntegratorBaseSparse
//                               (reference src/deformable/PS_VolumeConservingIntegrator.h:15-37,
//                                vegafem/integrator/integratorBase.h:107-205, integratorBaseSparse.h:45-84)
//   PS::FEM::Deformable     <-> class Deformable (reference src/deformable/Deformable.h:63-235): same method names,
//                               argument meaning and callback type; timestep() follows Deformable.cpp:318-420.
#pragma once
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../fembrain_hip.h"

namespace PS {
namespace FEM {

typedef unsigned int U32;
// same signature as the reference's FOnApplyDeformations (Deformable.h:46): borrowed pointer, dof = 3 * nodes
typedef void (*FOnApplyDeformations)(U32 dof, double* displacements);

struct vec3d {
  double x, y, z;
  vec3d(double x_ = 0, double y_ = 0, double z_ = 0) : x(x_), y(y_), z(z_) {}
};

class HipIntegrator {
 public:
  // constrainedDOFs: 0-indexed, ascending, copied (implicitNewmarkSparse.h:78-80)
  HipIntegrator(int numVertices, const double* restPositions, int numElements, const int* elements, int numConstrainedDOFs,
                const int* constrainedDOFs, double timestep = 0.0333, double dampingMassCoef = 0.0, double dampingStiffnessCoef = 0.01,
                double E = 1e7, double nu = 0.46, double density = 1000.0, int device = 0, int warp = 1, bool expectCuts = false)
      : r_(3 * numVertices), h_(nullptr) {
    // warp: the argument of CorotationalLinearFEMForceModel (corotationalLinearFEMForceModel.h:42): 1 = corotational (default,
    // what FemBrain runs), 0 = linear elasticity; 2 (exact tangent) is not offered -- its Keff is not symmetric
    if (warp != 0 && warp != 1) throw std::invalid_argument("warp must be 0 or 1");
    fb_fem_default_params(&prm_);
    prm_.linear = warp == 0 ? 1 : 0;
    prm_.E = E; prm_.nu = nu; prm_.rho = density;
    prm_.timestep = timestep; prm_.damping_mass = dampingMassCoef; prm_.damping_stiffness = dampingStiffnessCoef;
    prm_.device = device;
    prm_.expect_cuts = expectCuts ? 1 : 0;  // (a cuttable body: slack, node order and re-sync workspace at creation, fembrain_hip.h)
    check(fb_fem_create(&h_, numVertices, restPositions, numElements, elements, numConstrainedDOFs, constrainedDOFs, &prm_));
    std::memset(&info_, 0, sizeof info_);
  }
  // the tet mesh a polygonizer handle (GPUPoly::handle()) holds on the device becomes the FEM mesh, no host copy
  // (fb_fem_create_from_poly; field grid -> tets -> K0 on the device)
  HipIntegrator(fb_poly_t poly, int numConstrainedDOFs, const int* constrainedDOFs, double timestep = 0.0333, double dampingMassCoef = 0.0,
                double dampingStiffnessCoef = 0.01, double E = 1e7, double nu = 0.46, double density = 1000.0, int device = 0)
      : r_(0), h_(nullptr) {
    fb_fem_default_params(&prm_);
    prm_.E = E; prm_.nu = nu; prm_.rho = density;
    prm_.timestep = timestep; prm_.damping_mass = dampingMassCoef; prm_.damping_stiffness = dampingStiffnessCoef;
    prm_.device = device;
    check(fb_fem_create_from_poly(&h_, poly, numConstrainedDOFs, constrainedDOFs, &prm_));
    r_ = 3 * fb_fem_num_nodes(h_);
    std::memset(&info_, 0, sizeof info_);
  }
  ~HipIntegrator() { fb_fem_destroy(h_); }
  HipIntegrator(const HipIntegrator&) = delete;
  HipIntegrator& operator=(const HipIntegrator&) = delete;

  int Getr() const { return r_; }
  // 0 = ok, 1 = solver failed (the reference prints and exit(-1)s, PS_VolumeConservingIntegrator.cpp:203-209)
  int DoTimestep() {
    const int rc = fb_fem_step(h_, &info_);
    if (rc == FB_ESOLVER) return 1;
    check(rc);
    return 0;
  }
  void SetExternalForces(double* f) { check(fb_fem_set_external_forces(h_, f)); }
  void AddExternalForces(double* f) { check(fb_fem_add_external_forces(h_, f)); }
  void SetExternalForcesToZero() { check(fb_fem_set_external_forces_zero(h_)); }
  void SetUniformForce(int axis, double value) { check(fb_fem_set_uniform_force(h_, axis, value)); }
  void SetqState(const double* q, const double* qvel = nullptr, const double* qaccel = nullptr) { check(fb_fem_set_state(h_, q, qvel, qaccel)); }
  void GetqState(double* q, double* qvel = nullptr, double* qaccel = nullptr) { check(fb_fem_get_state(h_, q, qvel, qaccel)); }
  int SetState(double* q, double* qvel = nullptr) { check(fb_fem_set_state(h_, q, qvel, nullptr)); return 0; }
  // Getq / Getqvel / Getqaccel (integratorBase.h:139-141) hand out the integrator's own arrays in the reference; the state
  // lives on the device here, so these return a host mirror refreshed by the call (borrowed, valid until the next call)
  double* Getq() { mirror(); return mq_.data(); }
  double* Getqvel() { mirror(); return mv_.data(); }
  double* Getqaccel() { mirror(); return ma_.data(); }
  void ResetToRest() { check(fb_fem_reset(h_)); }
  void SetTimestep(double t) { prm_.timestep = t; check(fb_fem_set_timestep(h_, t)); }
  double GetTimestep() const { return prm_.timestep; }
  void SetDampingMassCoef(double c) { prm_.damping_mass = c; check(fb_fem_set_damping(h_, prm_.damping_mass, prm_.damping_stiffness)); }
  void SetDampingStiffnessCoef(double c) { prm_.damping_stiffness = c; check(fb_fem_set_damping(h_, prm_.damping_mass, prm_.damping_stiffness)); }
  void SetInternalForceScalingFactor(double f) { check(fb_fem_set_internal_force_scaling(h_, f)); }
  bool setConstrainedDOF(int num, int* arr) {
    if (num == 0 || arr == nullptr) return false;  // integratorBaseSparse.cpp:73-75
    check(fb_fem_set_constrained_dofs(h_, num, arr));
    return true;
  }
  // IntegratorBaseSparse::GetTotalMass / GetKineticEnergy (integratorBaseSparse.cpp:63-71): sum of the entries of the
  // inflated consistent mass matrix, and 1/2 qvel^T M qvel -- host arithmetic on the mass blocks (not on the step path)
  double GetTotalMass() {
    loadMass();
    double s = 0.0;
    for (size_t k = 0; k < mass_.size(); k++) s += mass_[k];
    return 3.0 * s;
  }
  double GetKineticEnergy() {
    loadMass();
    std::vector<double> v((size_t)r_);
    check(fb_fem_get_state(h_, nullptr, v.data(), nullptr));
    double e = 0.0;
    for (size_t a = 0; a + 1 < bptr_.size(); a++)
      for (int k = bptr_[a]; k < bptr_[a + 1]; k++) {
        const size_t b = (size_t)bcol_[k];
        e += mass_[k] * (v[3 * a] * v[3 * b] + v[3 * a + 1] * v[3 * b + 1] + v[3 * a + 2] * v[3 * b + 2]);
      }
    return 0.5 * e;
  }
  double GetForceAssemblyTime() const { return info_.assembly_seconds; }
  double GetSystemSolveTime() const { return info_.solve_seconds; }
  int GetLastIterations() const { return info_.cg_iterations; }
  int GetLastNewtonIterations() const { return info_.newton_iterations; }  // (Newmark: linear solves of the last step; 1 otherwise)
  int FloorCollision(double floorY, double restitution) {
    int n = 0;
    check(fb_fem_floor_collision(h_, floorY, restitution, &n));
    return n;
  }
  void RebuildElements() { check(fb_fem_rebuild_elements(h_)); }
  void Resync(int numVertices, const double* rest, int numElements, const int* elements, int nFixed, const int* fixed) {
    mass_.clear(); bptr_.clear(); bcol_.clear();
    check(fb_fem_resync(h_, numVertices, rest, numElements, elements, nFixed, fixed));
    r_ = 3 * numVertices;
  }
  // the same from a description of the change (fb_fem_resync_delta): the mesh stays on the device
  void ResyncDelta(int nRemoved, const int* removed, int nChanged, const int* changedIds, const int* changedNodes, int nAdded, const int* added, int nNewVertices,
                   const double* newRest, int nFixed, const int* fixed) {
    mass_.clear(); bptr_.clear(); bcol_.clear();
    check(fb_fem_resync_delta(h_, nRemoved, removed, nChanged, changedIds, changedNodes, nAdded, added, nNewVertices, newRest, nFixed, fixed));
    r_ = 3 * fb_fem_num_nodes(h_);
  }
  fb_fem_t handle() const { return h_; }
  static void check(int rc) {
    if (rc != FB_OK) throw std::runtime_error(std::string("fembrain_hip: ") + fb_last_error());
  }

 private:
  void loadMass() {
    if (!mass_.empty()) return;
    bptr_.resize((size_t)fb_fem_num_nodes(h_) + 1);
    bcol_.resize((size_t)fb_fem_num_blocks(h_));
    mass_.resize(bcol_.size());
    check(fb_fem_pattern(h_, bptr_.data(), bcol_.data()));
    check(fb_fem_mass(h_, mass_.data()));
  }
  std::vector<int> bptr_, bcol_;
  std::vector<double> mass_;
  int r_;
  void mirror() {
    mq_.resize((size_t)r_); mv_.resize((size_t)r_); ma_.resize((size_t)r_);
    check(fb_fem_get_state(h_, mq_.data(), mv_.data(), ma_.data()));
  }
  std::vector<double> mq_, mv_, ma_;
  fb_fem_t h_;
  fb_fem_params prm_;
  fb_step_info info_;
};

// ---- the narrower VegaFEM seams, for hosts that keep their own integrator -------------------------
// ForceModel (vegafem/forceModel/forceModel.h:42-67) over a HipIntegrator's handle: internal force and tangent stiffness
// at a displacement u.  The matrix comes as 3x3 blocks on the node-level pattern (bptr / bcol, ascending columns per row
// = the block rows of the reference's SparseMatrix, sparseMatrix.cpp:238-262), 9 row-major values per block.
class HipForceModel {
 public:
  explicit HipForceModel(HipIntegrator* integrator) : in_(integrator) {}
  int Getr() const { return in_->Getr(); }
  void GetInternalForce(double* u, double* internalForces) { HipIntegrator::check(fb_fem_assemble(in_->handle(), u, internalForces, nullptr)); }
  // GetTangentStiffnessMatrixTopology: callee-allocated in the reference; here the caller's vectors are filled
  void GetTangentStiffnessMatrixTopology(std::vector<int>& bptr, std::vector<int>& bcol) {
    bptr.resize((size_t)fb_fem_num_nodes(in_->handle()) + 1);
    bcol.resize((size_t)fb_fem_num_blocks(in_->handle()));
    HipIntegrator::check(fb_fem_pattern(in_->handle(), bptr.data(), bcol.data()));
  }
  void GetTangentStiffnessMatrix(double* u, std::vector<double>& blocks) {
    blocks.resize(9 * (size_t)fb_fem_num_blocks(in_->handle()));
    HipIntegrator::check(fb_fem_assemble(in_->handle(), u, nullptr, blocks.data()));
  }
  void GetForceAndMatrix(double* u, double* internalForces, std::vector<double>& blocks) {
    blocks.resize(9 * (size_t)fb_fem_num_blocks(in_->handle()));
    HipIntegrator::check(fb_fem_assemble(in_->handle(), u, internalForces, blocks.data()));
  }

 private:
  HipIntegrator* in_;
};

// LinearSolver (vegafem/sparseSolver/linearSolver.h:44-53) + CGSolver's Jacobi-PCG entry (CGSolver.cpp:129-190) on the
// system the handle assembled last (Keff of the current state): returns the reference's code, > 0 iterations when
// converged, < 0 when not.
class HipCGSolver {
 public:
  explicit HipCGSolver(HipIntegrator* integrator) : in_(integrator) {}
  int SolveLinearSystemWithJacobiPreconditioner(double* x, const double* b, double eps, int maxIterations) {
    int it = 0;
    HipIntegrator::check(fb_fem_pcg(in_->handle(), b, x, eps, maxIterations, &it));
    return it;
  }
  int SolveLinearSystem(double* x, const double* rhs) { return SolveLinearSystemWithJacobiPreconditioner(x, rhs, 1e-6, 10000) > 0 ? 0 : 1; }
  // CGSolver's C hook `typedef void (*blackBoxProductType)(const void* data, const double* x, double* Ax)` (CGSolver.h:65-66):
  // pass HipCGSolver::BlackBoxProduct with data = the HipIntegrator to run the reference's own CG on the device SpMV
  static void BlackBoxProduct(const void* data, const double* x, double* Ax) {
    HipIntegrator::check(fb_fem_spmv(static_cast<const HipIntegrator*>(data)->handle(), x, Ax));
  }

 private:
  HipIntegrator* in_;
};

class Deformable {
 public:
  Deformable(int numVertices, const double* restPositions, int numElements, const int* elements,
             const std::vector<int>& vFixedVertices = std::vector<int>(), int device = 0)
      : m_rest(restPositions, restPositions + 3 * (size_t)numVertices), m_elements(elements, elements + 4 * (size_t)numElements),
        m_vFixedVertices(vFixedVertices), m_device(device) {
    init();
    syncForceModel();
  }
  ~Deformable() { delete m_lpIntegrator; }

  // Deformable::timestep (Deformable.cpp:318-420)
  void timestep() {
    if (m_lpIntegrator == nullptr) return;
    const bool applyGravity = m_bApplyGravity && (m_ctCollided == 0);
    if (m_bHapticInProgress && !m_vHapticIndices.empty()) {
      std::fill(m_arrExtForces.begin(), m_arrExtForces.end(), 0.0);
      if (applyGravity)
        for (U32 i = 1; i < m_dof; i += 3) m_arrExtForces[i] += -10000.0;
      applyHapticForces();
      m_lpIntegrator->SetExternalForces(m_arrExtForces.data());
    } else if (applyGravity) {
      m_lpIntegrator->SetUniformForce(1, -10000.0);
    } else {
      m_lpIntegrator->SetExternalForcesToZero();
    }
    m_lpIntegrator->DoTimestep();
    if (m_hasFloor) m_ctCollided = (U32)m_lpIntegrator->FloorCollision(m_floorY, 0.4);
    if (m_fOnDeform) {
      m_lpIntegrator->GetqState(m_q.data(), m_qVel.data(), nullptr);
      m_fOnDeform(m_dof, m_q.data());
    }
    m_ctTimeStep++;
  }

  // rebuild after a topology change (Deformable.cpp:127-220)
  bool syncForceModel() {
    const int n = (int)(m_rest.size() / 3), m = (int)(m_elements.size() / 4);
    m_dof = 3 * n;
    FixedVerticesToFixedDOF(m_vFixedVertices, m_vFixedDofs);
    if (m_lpIntegrator) m_lpIntegrator->Resync(n, m_rest.data(), m, m_elements.data(), (int)m_vFixedDofs.size(), m_vFixedDofs.data());
    else m_lpIntegrator = new HipIntegrator(n, m_rest.data(), m, m_elements.data(), (int)m_vFixedDofs.size(), m_vFixedDofs.data(),
                                            m_timeStep, m_dampingMassCoeff, m_dampingStiffnessCoeff, 1e7, 0.46, 1000.0, m_device, 1,
                                            /* a Deformable is what FemBrain's scalpel cuts (CuttableMesh): */ true);
    m_q.assign(m_dof, 0.0); m_qVel.assign(m_dof, 0.0); m_arrExtForces.assign(m_dof, 0.0);
    m_bptr.clear(); m_bcol.clear();
    m_restVolume = -1.0;
    return true;
  }
  // The same after a cut that the host can describe (CuttableMesh::cut erases the cut cells -- VolMesh.cpp:630 -- appends their pieces
  // and the new nodes -- :1083-1088 -- and re-points cells in place -- :1630-1650): `removed` ascending ids of the current element list,
  // `changedIds` ascending with their new nodes, `added` and `newRest` appended.  The host copy of the mesh follows.
  bool syncForceModelDelta(const std::vector<int>& removed, const std::vector<int>& changedIds, const std::vector<int>& changedNodes, const std::vector<int>& added,
                           const std::vector<double>& newRest) {
    if (!m_lpIntegrator) return syncForceModel();
    // The DEVICE first: ResyncDelta refuses bad input before anything changes and throws; the host copy of the mesh is edited
    // only after it has returned, so host and device never disagree.  Ids are range-checked here too: they index the host arrays below.
    const size_t n_el = m_elements.size() / 4;
    for (size_t k = 0; k < changedIds.size(); k++)
      if (changedIds[k] < 0 || (size_t)changedIds[k] >= n_el) throw std::runtime_error("syncForceModelDelta: changed element id out of range");
    for (size_t k = 0; k < removed.size(); k++)
      if (removed[k] < 0 || (size_t)removed[k] >= n_el || (k && removed[k] <= removed[k - 1])) throw std::runtime_error("syncForceModelDelta: removed ids must ascend inside the element list");
    if (changedNodes.size() != 4 * changedIds.size() || added.size() % 4 != 0 || newRest.size() % 3 != 0) throw std::runtime_error("syncForceModelDelta: array sizes do not match");
    std::vector<int> fixedDofs;
    FixedVerticesToFixedDOF(m_vFixedVertices, fixedDofs);
    m_lpIntegrator->ResyncDelta((int)removed.size(), removed.data(), (int)changedIds.size(), changedIds.data(), changedNodes.data(), (int)(added.size() / 4), added.data(),
                                (int)(newRest.size() / 3), newRest.data(), (int)fixedDofs.size(), fixedDofs.data());
    for (size_t k = 0; k < changedIds.size(); k++)
      for (int c = 0; c < 4; c++) m_elements[4 * (size_t)changedIds[k] + c] = changedNodes[4 * k + c];
    if (!removed.empty()) {
      std::vector<int> kept;
      kept.reserve(m_elements.size());
      size_t r = 0;
      for (size_t e = 0; e < n_el; e++) {
        if (r < removed.size() && (size_t)removed[r] == e) { r++; continue; }
        kept.insert(kept.end(), m_elements.begin() + 4 * e, m_elements.begin() + 4 * e + 4);
      }
      m_elements.swap(kept);
    }
    m_elements.insert(m_elements.end(), added.begin(), added.end());
    m_rest.insert(m_rest.end(), newRest.begin(), newRest.end());
    m_dof = (U32)m_rest.size();
    m_vFixedDofs.swap(fixedDofs);
    m_q.assign(m_dof, 0.0); m_qVel.assign(m_dof, 0.0); m_arrExtForces.assign(m_dof, 0.0);
    m_bptr.clear(); m_bcol.clear();
    m_restVolume = -1.0;
    return true;
  }
  void setMesh(int numVertices, const double* rest, int numElements, const int* elements) {
    m_rest.assign(rest, rest + 3 * (size_t)numVertices);
    m_elements.assign(elements, elements + 4 * (size_t)numElements);
  }

  // Deformable::FixedVerticesToFixedDOF (Deformable.cpp:294-314)
  static int FixedVerticesToFixedDOF(std::vector<int>& arrInFixedVertices, std::vector<int>& arrOutFixedDOF) {
    arrOutFixedDOF.clear();
    if (arrInFixedVertices.empty()) return 0;
    std::sort(arrInFixedVertices.begin(), arrInFixedVertices.end());
    arrOutFixedDOF.resize(arrInFixedVertices.size() * 3);
    for (size_t i = 0; i < arrInFixedVertices.size(); i++)
      for (int k = 0; k < 3; k++) arrOutFixedDOF[3 * i + k] = arrInFixedVertices[i] * 3 + k;
    return (int)arrOutFixedDOF.size();
  }

  bool addFixedVertex(int index) {
    if (std::find(m_vFixedVertices.begin(), m_vFixedVertices.end(), index) != m_vFixedVertices.end()) return false;
    m_vFixedVertices.push_back(index);
    return updateFixedVertices();
  }
  bool removeFixedVertex(int index) {
    std::vector<int>::iterator it = std::find(m_vFixedVertices.begin(), m_vFixedVertices.end(), index);
    if (it == m_vFixedVertices.end()) return false;
    m_vFixedVertices.erase(it);
    return updateFixedVertices();
  }
  bool setFixedVertices(const std::vector<int>& v) { m_vFixedVertices = v; return updateFixedVertices(); }
  int getFixedVertices(std::vector<int>& v) { v = m_vFixedVertices; return (int)v.size(); }
  bool updateFixedVertices() {
    FixedVerticesToFixedDOF(m_vFixedVertices, m_vFixedDofs);
    return m_lpIntegrator->setConstrainedDOF((int)m_vFixedDofs.size(), m_vFixedDofs.data());
  }

  // Deformable::applyHapticForces (Deformable.cpp:634-706): each haptic force acts on its vertex and, with the linear
  // fall-off (size - j)/size, on the vertices first reached in ring j of a breadth-first walk over mesh edges.  Vertex
  // neighbours are the off-diagonal columns of the stiffness pattern (vertices sharing a tet edge).
  bool applyHapticForces() {
    if (m_vHapticIndices.empty() || !m_bHapticInProgress) return false;
    if (m_bptr.empty()) {
      m_bptr.resize(m_rest.size() / 3 + 1);
      m_bcol.resize(fb_fem_num_blocks(m_lpIntegrator->handle()));
      if (fb_fem_pattern(m_lpIntegrator->handle(), m_bptr.data(), m_bcol.data()) != FB_OK) return false;
    }
    for (size_t i = 0; i < m_vHapticIndices.size(); i++) {
      const int v = m_vHapticIndices[i];
      m_arrExtForces[3 * v + 0] += m_vHapticForces[i].x;
      m_arrExtForces[3 * v + 1] += m_vHapticForces[i].y;
      m_arrExtForces[3 * v + 2] += m_vHapticForces[i].z;
    }
    for (size_t iv = 0; iv < m_vHapticIndices.size(); iv++) {
      std::vector<char> affected(m_rest.size() / 3, 0);
      std::vector<int> last(1, m_vHapticIndices[iv]);
      affected[m_vHapticIndices[iv]] = 1;
      const vec3d f = m_vHapticForces[iv];
      for (int j = 1; j < m_hapticForceNeighorhoodSize; j++) {
        const double mag = 1.0 * (m_hapticForceNeighorhoodSize - j) / static_cast<double>(m_hapticForceNeighorhoodSize);
        std::vector<int> fresh;
        for (size_t a = 0; a < last.size(); a++)
          for (int p = m_bptr[last[a]]; p < m_bptr[last[a] + 1]; p++) {
            const int nb = m_bcol[p];
            if (!affected[nb]) { affected[nb] = 2; fresh.push_back(nb); }  // 2 = discovered in this ring
          }
        std::sort(fresh.begin(), fresh.end());
        for (size_t a = 0; a < fresh.size(); a++) {
          m_arrExtForces[3 * fresh[a] + 0] += mag * f.x;
          m_arrExtForces[3 * fresh[a] + 1] += mag * f.y;
          m_arrExtForces[3 * fresh[a] + 2] += mag * f.z;
          affected[fresh[a]] = 1;
        }
        last.swap(fresh);
      }
    }
    return true;
  }
  int getHapticForceRadius() const { return m_hapticForceNeighorhoodSize; }
  void setHapticForceRadius(int radius) { m_hapticForceNeighorhoodSize = radius; }
  void setPulledVertex(int index) { m_idxPulledVertex = index; }
  bool hapticStart(int index) { m_idxPulledVertex = index; m_bHapticInProgress = true; m_vHapticIndices.clear(); return true; }
  void hapticEnd() { m_bHapticInProgress = false; m_idxPulledVertex = -1; m_vHapticIndices.clear(); }
  bool isHapticInProgress() const { return m_bHapticInProgress; }
  void hapticSetCurrentForces(const std::vector<int>& indices, const std::vector<vec3d>& forces) { m_vHapticIndices = indices; m_vHapticForces = forces; }
  void setDampingStiffnessCoeff(double s) { m_dampingStiffnessCoeff = s; m_lpIntegrator->SetDampingStiffnessCoef(s); }
  double getDampingStiffnessCoeff() const { return m_dampingStiffnessCoeff; }
  void setDampingMassCoeff(double m) { m_dampingMassCoeff = m; m_lpIntegrator->SetDampingMassCoef(m); }
  double getDampingMassCoeff() const { return m_dampingMassCoeff; }
  void setGravity(bool g) { m_bApplyGravity = g; }
  bool getGravity() const { return m_bApplyGravity; }
  void setFloor(double y) { m_hasFloor = true; m_floorY = y; }  // stands in for setCollisionObject(SGNode*): the floor plane height
  // Deformable::collisionDetect (Deformable.cpp:541-600) against the floor plane set with setFloor: vertices at or below it are
  // put back on it and their velocity reflected (the device pass timestep() also runs); true if any vertex was touched
  bool collisionDetect() {
    if (!m_hasFloor) return false;
    m_ctCollided = (U32)m_lpIntegrator->FloorCollision(m_floorY, 0.4);
    return m_ctCollided > 0;
  }
  // Deformable::statFillRecord (Deformable.cpp:225-258) without the sqlite logger: the same fields in a plain record
  struct StatRecord {
    U32 ctElements, ctVertices;
    double restVolume, totalVolume, youngModulo, poissonRatio;
    const char *xpElementType, *xpForceModel, *xpIntegrator;
    std::string xpModelName;
  };
  void statFillRecord(StatRecord& rec) {
    rec.ctElements = countCells(); rec.ctVertices = countNodes();
    (void)isVolumeChanged();  // fills the rest volume on first use
    rec.restVolume = m_restVolume; rec.totalVolume = computeVolume();
    rec.youngModulo = 0.0; rec.poissonRatio = 0.0;  // as the reference leaves them
    rec.xpElementType = "TET"; rec.xpForceModel = "COROTATIONAL LINEAR FEM"; rec.xpIntegrator = "JACOBI PRECONDITIONED CG";
    rec.xpModelName = m_strModelName;
  }
  void resetDeformations() { m_lpIntegrator->ResetToRest(); m_vHapticForces.clear(); }
  void setDeformCallback(FOnApplyDeformations fOnDeform) { m_fOnDeform = fOnDeform; }
  double getSolverTime() const { return m_lpIntegrator->GetSystemSolveTime(); }
  // Deformable::pickVertex (Deformable.cpp:422-428): closest vertex of the displaced mesh to a world position
  int pickVertex(const vec3d& wpos, vec3d& vertex) {
    m_lpIntegrator->GetqState(m_q.data(), nullptr, nullptr);
    int best = -1;
    double bd = 0.0;
    for (size_t i = 0; i < m_rest.size() / 3; i++) {
      const double dx = m_rest[3 * i] + m_q[3 * i] - wpos.x, dy = m_rest[3 * i + 1] + m_q[3 * i + 1] - wpos.y, dz = m_rest[3 * i + 2] + m_q[3 * i + 2] - wpos.z;
      const double d = dx * dx + dy * dy + dz * dz;
      if (best < 0 || d < bd) { best = (int)i; bd = d; }
    }
    if (best >= 0) { vertex.x = m_rest[3 * (size_t)best] + m_q[3 * (size_t)best]; vertex.y = m_rest[3 * (size_t)best + 1] + m_q[3 * (size_t)best + 1]; vertex.z = m_rest[3 * (size_t)best + 2] + m_q[3 * (size_t)best + 2]; }
    return best;
  }
  // Deformable::pickVertices (Deformable.cpp:430-448)
  int pickVertices(const vec3d& boxLo, const vec3d& boxHi, std::vector<vec3d>& arrFoundCoords, std::vector<int>& arrFoundIndices) {
    m_lpIntegrator->GetqState(m_q.data(), nullptr, nullptr);
    arrFoundCoords.clear(); arrFoundIndices.clear();
    for (size_t i = 0; i < m_rest.size() / 3; i++) {
      vec3d v = {m_rest[3 * i] + m_q[3 * i], m_rest[3 * i + 1] + m_q[3 * i + 1], m_rest[3 * i + 2] + m_q[3 * i + 2]};
      if (v.x >= boxLo.x && v.x <= boxHi.x && v.y >= boxLo.y && v.y <= boxHi.y && v.z >= boxLo.z && v.z <= boxHi.z) {
        arrFoundCoords.push_back(v);
        arrFoundIndices.push_back((int)i);
      }
    }
    return (int)arrFoundCoords.size();
  }
  // Deformable::hapticStart(const vec3d&) (Deformable.cpp:519-532): refuses a clamped vertex
  bool hapticStart(const vec3d& wpos) {
    vec3d vertex;
    const int idx = pickVertex(wpos, vertex);
    for (size_t i = 0; i < m_vFixedVertices.size(); i++)
      if (m_vFixedVertices[i] == idx) { m_idxPulledVertex = idx; return false; }
    return hapticStart(idx);
  }
  // Deformable::isVolumeChanged (Deformable.h:128); the rest volume is taken at the first call after a (re)build
  bool isVolumeChanged() {
    if (m_restVolume < 0.0) {
      std::vector<double> keep(m_q);
      std::fill(m_q.begin(), m_q.end(), 0.0);
      m_restVolume = volumeOf(m_q);
      m_q.swap(keep);
    }
    return std::abs(computeVolume() - m_restVolume) > 0.0001;
  }
  std::string getModelName() const { return m_strModelName; }
  void setModelName(const std::string& name) { m_strModelName = name; }
  U32 getDof() const { return m_dof; }
  U32 getCollidedCount() const { return m_ctCollided; }
  HipIntegrator* integrator() { return m_lpIntegrator; }
  // what getVolMesh()->countNodes() / countCells() / const_nodeAt(i).pos / const_cellAt(e).nodes give the cutting tool
  // (Cutting.cpp:96-147): the current (displaced) node positions and the element list
  U32 countNodes() const { return (U32)(m_rest.size() / 3); }
  U32 countCells() const { return (U32)(m_elements.size() / 4); }
  const std::vector<int>& cells() const { return m_elements; }
  std::vector<double> currentPositions() {
    if (m_lpIntegrator) m_lpIntegrator->GetqState(m_q.data(), nullptr, nullptr);
    std::vector<double> p(m_rest);
    for (size_t i = 0; i < p.size() && i < m_q.size(); i++) p[i] += m_q[i];
    return p;
  }
  // Deformable::computeVolume (Deformable.cpp:260-279) on the current displaced positions
  double computeVolume(double* arrStore = nullptr, U32 count = 0) {
    m_lpIntegrator->GetqState(m_q.data(), nullptr, nullptr);
    return volumeOf(m_q, arrStore, count);
  }

 private:
  double volumeOf(const std::vector<double>& q, double* arrStore = nullptr, U32 count = 0) const {
    const U32 m = (U32)(m_elements.size() / 4);
    const bool store = arrStore != nullptr && count == m;
    double vol = 0.0;
    for (U32 e = 0; e < m; e++) {
      double p[4][3];
      for (int k = 0; k < 4; k++)
        for (int d = 0; d < 3; d++) p[k][d] = m_rest[3 * (size_t)m_elements[4 * e + k] + d] + q[3 * (size_t)m_elements[4 * e + k] + d];
      const double u[3] = {p[0][0] - p[3][0], p[0][1] - p[3][1], p[0][2] - p[3][2]}, v[3] = {p[1][0] - p[3][0], p[1][1] - p[3][1], p[1][2] - p[3][2]},
                   w[3] = {p[2][0] - p[3][0], p[2][1] - p[3][1], p[2][2] - p[3][2]};
      const double cur = std::abs(u[0] * (v[1] * w[2] - v[2] * w[1]) + u[1] * (v[2] * w[0] - v[0] * w[2]) + u[2] * (v[0] * w[1] - v[1] * w[0])) / 6.0;
      if (store) arrStore[e] = cur;
      vol += cur;
    }
    return vol;
  }
  void init() {  // Deformable::init (Deformable.cpp:85-124)
    m_ctCollided = 0; m_fOnDeform = nullptr; m_idxPulledVertex = -1; m_bHapticInProgress = false;
    m_dampingMassCoeff = 0.0; m_dampingStiffnessCoeff = 0.01; m_timeStep = 0.0333; m_ctTimeStep = 0;
    m_hapticForceNeighorhoodSize = 5;  // DEFAULT_FORCE_NEIGHBORHOOD_SIZE, Deformable.h:41
    m_bApplyGravity = true;  // left uninitialised by the reference's init(); true is what its .sim files set
    m_lpIntegrator = nullptr; m_hasFloor = false; m_floorY = 0.0; m_dof = 0; m_restVolume = -1.0;
  }
  std::vector<double> m_rest;
  std::vector<int> m_elements;
  std::vector<int> m_vFixedVertices, m_vFixedDofs, m_vHapticIndices, m_bptr, m_bcol;
  std::vector<vec3d> m_vHapticForces;
  std::vector<double> m_q, m_qVel, m_arrExtForces;
  HipIntegrator* m_lpIntegrator;
  FOnApplyDeformations m_fOnDeform;
  U32 m_dof, m_ctCollided, m_ctTimeStep;
  int m_idxPulledVertex, m_device, m_hapticForceNeighorhoodSize;
  bool m_bHapticInProgress, m_bApplyGravity, m_hasFloor;
  double m_dampingMassCoeff, m_dampingStiffnessCoeff, m_timeStep, m_floorY, m_restVolume;
  std::string m_strModelName;
};

}  // namespace FEM
}  // namespace PS
