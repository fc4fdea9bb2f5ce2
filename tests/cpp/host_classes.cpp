// Host program over the C++ class surfaces (include/fembrain/Deformable.h, GPUPoly.h): a truth cube stepped through
// PS::FEM::Deformable with a deformation callback, then sphere.blob polygonized through PS::SKETCH::GPUPoly and the
// resulting tet mesh fed back into a Deformable -- the reference's main.cpp flow (src/main.cpp:782-886) without GL.
// Prints one KEY=VALUE line per fact for tests/test_cpp_host.py.
#include <cmath>
#include <cstdio>
#include <vector>

#include "fembrain/Deformable.h"
#include "fembrain/Cutting.h"
#include "fembrain/GPUPoly.h"

static unsigned g_calls = 0;
static double g_maxq = 0;
static void onDeform(PS::FEM::U32 dof, double* q) {
  g_calls++;
  g_maxq = 0;
  for (unsigned i = 0; i < dof; i++) g_maxq = std::fmax(g_maxq, std::fabs(q[i]));
}

int main() {
  // CreateTruthCube(5,5,5,0.1), src/deformable/VolMeshSamples.cpp:67-130
  const int n = 5;
  std::vector<double> v;
  std::vector<int> t;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) for (int k = 0; k < n; k++) {
    v.push_back((-n / 2.0 + i) * 0.1); v.push_back(j * 0.1); v.push_back((-n / 2.0 + k) * 0.1);
  }
  for (int i = 0; i < n - 1; i++) for (int j = 0; j < n - 1; j++) for (int k = 0; k < n - 1; k++) {
    int c[8];
    for (int q = 0; q < 8; q++) c[q] = (i + ((q >> 2) & 1)) * n * n + (j + ((q >> 1) & 1)) * n + k + (q & 1);
    const int pat[6][4] = {{0, 2, 4, 1}, {6, 2, 1, 4}, {6, 2, 3, 1}, {6, 4, 1, 5}, {6, 1, 3, 5}, {6, 3, 7, 5}};
    for (int a = 0; a < 6; a++) for (int b = 0; b < 4; b++) t.push_back(c[pat[a][b]]);
  }
  std::vector<int> fixed;
  for (int a = 0; a < n * n; a++) fixed.push_back(a);
  PS::FEM::Deformable d(n * n * n, v.data(), (int)t.size() / 4, t.data(), fixed);
  d.setDeformCallback(onDeform);
  const double vol0 = d.computeVolume();
  d.timestep();
  d.timestep();
  std::printf("CUBE_DOF=%u\nCUBE_CALLBACKS=%u\nCUBE_MAXQ=%.9g\nCUBE_VOL0=%.9g\nCUBE_VOL=%.9g\nCUBE_ITERS=%d\nCUBE_SOLVE_S=%.6g\n", d.getDof(), g_calls, g_maxq,
              vol0, d.computeVolume(), d.integrator()->GetLastIterations(), d.getSolverTime());

  // the narrower VegaFEM seams on the same handle: force model, black-box product, CG solver
  {
    PS::FEM::HipForceModel fm(d.integrator());
    PS::FEM::HipCGSolver cg(d.integrator());
    const int r = fm.Getr();
    std::vector<double> u(r, 0.0), f(r, 0.0), blocks, x(r, 0.0), ax(r, 0.0), b(r, 0.0);
    std::vector<int> bptr, bcol;
    for (int i = 0; i < r; i++) u[i] = 1e-3 * std::sin(0.37 * i);
    fm.GetTangentStiffnessMatrixTopology(bptr, bcol);
    fm.GetForceAndMatrix(u.data(), f.data(), blocks);
    double fn = 0, kn = 0;
    for (int i = 0; i < r; i++) fn += f[i] * f[i];
    for (size_t i = 0; i < blocks.size(); i++) kn += blocks[i] * blocks[i];
    for (int i = 0; i < r; i++) b[i] = std::cos(0.11 * i);
    const int it = cg.SolveLinearSystemWithJacobiPreconditioner(x.data(), b.data(), 1e-8, 5000);   // Keff of the last step
    PS::FEM::HipCGSolver::BlackBoxProduct(d.integrator(), x.data(), ax.data());
    double res = 0, bn = 0;
    for (int i = 0; i < r; i++) { res += (ax[i] - b[i]) * (ax[i] - b[i]); bn += b[i] * b[i]; }
    std::printf("TOTAL_MASS=%.12g\nKINETIC=%.12g\n", d.integrator()->GetTotalMass(), d.integrator()->GetKineticEnergy());
    std::printf("SEAM_BLOCKS=%zu\nSEAM_FNORM=%.12g\nSEAM_KNORM=%.12g\nSEAM_CG_ITERS=%d\nSEAM_RESIDUAL=%.3g\n", bcol.size(), std::sqrt(fn), std::sqrt(kn), it,
                std::sqrt(res / bn));
  }

  // picking / volume check on the displaced mesh
  PS::FEM::vec3d far = {10.0, 0.2, 10.0}, hit;
  const int picked = d.pickVertex(far, hit);            // nearest vertex to a far +x,+z point: the (n-1, j, n-1) corner column
  PS::FEM::vec3d boxLo = {-1.0, -1.0, -1.0}, boxHi = {-0.24, 1.0, 1.0};
  std::vector<PS::FEM::vec3d> found;
  std::vector<int> foundIdx;
  const int nbox = d.pickVertices(boxLo, boxHi, found, foundIdx);   // the clamped plane i = 0 (x = -0.25)
  std::printf("PICKED=%d\nPICK_BOX=%d\nVOL_CHANGED=%d\nHAPTIC_FIXED=%d\n", picked, nbox, d.isVolumeChanged() ? 1 : 0,
              d.hapticStart(PS::FEM::vec3d{-0.25, 0.0, -0.25}) ? 1 : 0);
  d.hapticEnd();
  {
    PS::FEM::Deformable::StatRecord rec;
    d.statFillRecord(rec);
    std::printf("STAT=%u,%u,%s,%s\nSTAT_VOL=%.9g\nCOLLIDE_NOFLOOR=%d\n", rec.ctElements, rec.ctVertices, rec.xpElementType, rec.xpIntegrator, rec.restVolume,
                d.collisionDetect() ? 1 : 0);
  }

  // sphere.blob -> GPUPoly -> tet mesh -> Deformable
  PS::SKETCH::LinearBlobTreeData blob;
  const float hdr[12] = {-0.5f, -0.5f, -0.5f, 1, 0.5f, 0.5f, 0.5f, 1, 1, 0, 1, 65535.0f};
  blob.header.assign(hdr, hdr + 12);
  blob.prims.assign(20, 0.0f);
  const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  blob.mtx.assign(ident, ident + 12);
  PS::SKETCH::GPUPoly poly(blob);
  poly.setCellSize(0.1f);
  const int ok = poly.run();
  const int ntets = poly.runTetrahedralizer();
  PS::SKETCH::U32 nv, nt;
  std::vector<float> xyz;
  std::vector<PS::SKETCH::U32> el;
  poly.readbackTetMesh(nv, xyz, nt, el);
  std::printf("POLY_RUN=%d\nPOLY_TETS=%d\nPOLY_VERTS=%u\n", ok, ntets, nv);
  PS::SKETCH::U32 sv, st;
  std::vector<float> sxyz, snrm, moved;
  std::vector<PS::SKETCH::U32> sel;
  poly.readbackMeshV3T3(sv, sxyz, st, sel);
  poly.readBackNormals(sv, sxyz, snrm);
  std::vector<double> shift(3 * (size_t)sv, 0.25);
  const bool applied = poly.applyFemDisplacements(3 * sv, shift.data(), &moved);
  const float lower[3] = {hdr[0], hdr[1], hdr[2]};
  const std::vector<float> vox = poly.surfaceVoxels(lower);
  std::printf("SURF_VOXELS=%zu\nSURF_VOXELS_COUNT=%u\n", vox.size() / 3, poly.countSurfaceVoxels());
  std::printf("SURF_VERTS=%u\nSURF_TRIS=%u\nSURF_APPLIED=%d\nSURF_SHIFT=%.9g\n", sv, st, applied ? 1 : 0, sv ? moved[0] - sxyz[0] : 0.0f);
  PS::SKETCH::FieldComputer fc(blob);
  std::printf("FIELD_025=%.9g\nGRID_POINTS=%d\n", fc.field(0.25f, 0, 0), fc.fieldsForVoxelGrid(0.1f));
  std::vector<double> xv(xyz.begin(), xyz.end());
  std::vector<int> ev(el.begin(), el.end());
  std::vector<int> low;
  for (unsigned i = 0; i < nv; i++) if (xyz[3 * i + 1] < -0.35f) low.push_back((int)i);  // BASELINE config 1 clamp
  PS::FEM::Deformable ball((int)nv, xv.data(), (int)nt, ev.data(), low);
  ball.timestep();
  std::printf("BALL_FIXED=%zu\nBALL_ITERS=%d\n", low.size(), ball.integrator()->GetLastIterations());
  {  // the same ball with the mesh handed over on the device (no host copy): one step must give the same iteration count
    PS::FEM::HipIntegrator direct(poly.handle(), 0, nullptr);
    std::vector<double> f((size_t)direct.Getr(), 0.0);
    for (size_t i = 1; i < f.size(); i += 3) f[i] = -10.0;
    direct.SetExternalForces(f.data());
    PS::FEM::HipIntegrator staged((int)nv, xv.data(), (int)nt, ev.data(), 0, nullptr);
    staged.SetExternalForces(f.data());
    const int a = direct.DoTimestep(), b = staged.DoTimestep();
    std::printf("DIRECT_R=%d\nDIRECT_SAME=%d\n", direct.Getr(), (a == b && direct.GetLastIterations() == staged.GetLastIterations()) ? 1 : 0);
  }
  {  // the field path as two "ranks" run it (one after the other here): the pieces put together are the mesh above
    const int dims[3] = {12, 12, 12};
    std::vector<int> counts(2, 0);
    size_t verts = 0, tets = 0;
    bool same = true;
    for (int pass = 0; pass < 2; pass++)      // pass 0 learns the counts an all-gather would deliver
      for (int r = 0; r < 2; r++) {
        PS::SKETCH::GPUPoly part(blob);
        part.setCellSize(0.1f);
        PS::SKETCH::U32 pv, pt;
        std::vector<float> px;
        std::vector<PS::SKETCH::U32> pe;
        int mine = 0;
        part.runTetrahedralizerSlab(hdr, dims, r, 2, [&](int n) { mine = n; return counts; }, pv, px, pt, pe);
        if (pass == 0) { counts[r] = mine; continue; }
        for (size_t i = 0; i < px.size(); i++) same = same && px[i] == xyz[3 * verts + i];
        for (size_t i = 0; i < pe.size(); i++) same = same && pe[i] == el[4 * tets + i];
        verts += pv; tets += pt;
      }
    std::printf("SLAB_VERTS=%zu\nSLAB_TETS=%zu\nSLAB_SAME=%d\n", verts, tets, same ? 1 : 0);
  }
  {  // Deformable::syncForceModelDelta: the cube with its first two elements cut out, split on a new centre node, one changed in place --
     // against a Deformable made from the resulting mesh: same first step
    std::vector<double> mid(3, 0.0);
    for (int c = 0; c < 4; c++) for (int k = 0; k < 3; k++) mid[k] += 0.25 * v[3 * (size_t)t[c] + k];
    const int m = n * n * n;
    std::vector<int> removed = {0}, changedIds = {1}, changedNodes = {t[4], t[6], t[5], t[7]};
    std::vector<int> added = {t[0], t[1], t[2], m, m, t[1], t[2], t[3], t[0], m, t[2], t[3], t[0], t[1], m, t[3]};
    PS::FEM::Deformable dd(n * n * n, v.data(), (int)t.size() / 4, t.data(), fixed);
    dd.timestep();
    dd.syncForceModelDelta(removed, changedIds, changedNodes, added, mid);
    std::vector<double> v2 = v;
    v2.insert(v2.end(), mid.begin(), mid.end());
    std::vector<int> t2(t.begin() + 4, t.end());
    for (int c = 0; c < 4; c++) t2[c] = changedNodes[c];
    t2.insert(t2.end(), added.begin(), added.end());
    PS::FEM::Deformable fresh(m + 1, v2.data(), (int)t2.size() / 4, t2.data(), fixed);
    dd.timestep();
    fresh.timestep();
    const std::vector<double> a = dd.currentPositions(), b = fresh.currentPositions();
    bool same = a.size() == b.size() && dd.countCells() == fresh.countCells();
    for (size_t i = 0; same && i < a.size(); i++) same = a[i] == b[i];
    std::printf("DELTA_PATH=%d\nDELTA_NODES=%u\nDELTA_CELLS=%u\nDELTA_SAME=%d\n", fb_fem_resync_path(dd.integrator()->handle()), dd.countNodes(), dd.countCells(), same ? 1 : 0);
  }
  // the cutting tool on the ball's current (deformed) mesh: a vertical needle through it, then a blade swept along x
  PS::FEM::Cutting cut(&ball);
  float known[4];
  cut.computeFaceSegmentIntersectionTest(known);
  std::printf("CUT_KNOWN=%g,%g,%g,%g\n", known[0], known[1], known[2], known[3]);
  const int nface = cut.computeFaceIntersections(PS::FEM::vec3d(0.013, -2.0, 0.021), PS::FEM::vec3d(0.013, 2.0, 0.021));
  std::vector<PS::FEM::U32> ids, flags;
  std::vector<float> pts, all;
  cut.readHits(FB_CUT_FACES, ids, pts);
  cut.readFacePoints(flags, all);
  std::printf("CUT_FACES=%d\nCUT_FACE_IDS=%zu\nCUT_FACE_FLAGS=%zu\n", nface, ids.size(), flags.size());
  cut.performCut(PS::FEM::vec3d(-0.2, 0.05, -1.0), PS::FEM::vec3d(-0.2, 0.05, 1.0));
  cut.performCut(PS::FEM::vec3d(-0.1, 0.05, -1.0), PS::FEM::vec3d(-0.1, 0.05, 1.0));
  const bool before = cut.isSweptQuadValid();
  cut.performCut(PS::FEM::vec3d(0.1, 0.05, -1.0), PS::FEM::vec3d(0.1, 0.05, 1.0));
  std::printf("CUT_QUAD=%d%d\nCUT_EDGES=%u\nCUT_QUAD_X=%g\n", before ? 1 : 0, cut.isSweptQuadValid() ? 1 : 0, cut.countEdgePoints(), cut.sweptQuad()[2].x);
  {
    std::vector<double> cur = ball.currentPositions();
    std::printf("CUT_NODES=%u\nCUT_CELLS=%u\nCUT_POS0=%.17g\n", ball.countNodes(), ball.countCells(), cur[1]);
  }
  return 0;
}
