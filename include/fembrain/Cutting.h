// PS::FEM::Cutting (reference src/deformable/Cutting.h:34-111, Cutting.cpp) over the C ABI: the scalpel / tet-mesh
// intersection passes (the four kernels of data/opencl/Cutting.cl) and performCut's swept-quad bookkeeping.  Header-only;
// link with -lfembrain_hip.  The SGMesh base, draw() and the GL point buffers of the original stay in the host
// application: read the face / edge points back with readFacePoints / readEdgePoints and hand them to the renderer.
#pragma once
#include <cmath>
#include <stdexcept>
#include <vector>

#include "../fembrain_hip.h"
#include "Deformable.h"

namespace PS {
namespace FEM {

class Cutting {
 public:
  // Cutting::Cutting(Deformable*) + setup() + createMemBuffers() (Cutting.cpp:24-167)
  explicit Cutting(Deformable* lpDeformableModel, int device = 0) : m_lpDeformable(lpDeformableModel), h_(nullptr), m_device(device) { init(); createMemBuffers(); }
  // the same from bare arrays
  Cutting(U32 ctVertices, const double* xyz, U32 ctTets, const U32* tets, int device = 0) : m_lpDeformable(nullptr), h_(nullptr), m_device(device) {
    init();
    m_vMeshVertices.assign(xyz, xyz + 3 * (size_t)ctVertices);
    m_vMeshTets.assign(tets, tets + 4 * (size_t)ctTets);
    upload();
  }
  ~Cutting() { cleanup(); }
  Cutting(const Cutting&) = delete;
  Cutting& operator=(const Cutting&) = delete;

  void cleanup() { cleanupMemBuffers(); }
  void cleanupMemBuffers() {
    fb_cut_destroy(h_);
    h_ = nullptr;
    m_isMemBuffersLoaded = false;
  }
  // re-reads the deformable's current node positions and elements (call after a step or a topology change)
  bool createMemBuffers() {
    if (!m_lpDeformable) return false;
    const std::vector<int>& cells = m_lpDeformable->cells();
    m_vMeshTets.assign(cells.begin(), cells.end());
    m_vMeshVertices = m_lpDeformable->currentPositions();
    return upload();
  }

  // Cutting::performCut(edge0, edge1) (Cutting.cpp:499-535)
  int performCut(const vec3d& edge0, const vec3d& edge1) {
    const double minSweptLength = 0.01;
    m_isSweptQuadValid = false;
    m_sweptQuad[0] = edge0;
    m_sweptQuad[1] = edge1;
    if (m_vCuttingPathEdge0.size() > 1) {
      for (int i = (int)m_vCuttingPathEdge0.size() - 1; i >= 0; i--) {
        const vec3d& p = m_vCuttingPathEdge0[i];
        const double d = std::sqrt((edge0.x - p.x) * (edge0.x - p.x) + (edge0.y - p.y) * (edge0.y - p.y) + (edge0.z - p.z) * (edge0.z - p.z));
        if (d >= minSweptLength) {
          m_sweptQuad[2] = m_vCuttingPathEdge0[i];
          m_sweptQuad[3] = m_vCuttingPathEdge1[i];
          m_isSweptQuadValid = true;
          break;
        }
      }
    }
    const size_t maxNodes = 512;
    m_vCuttingPathEdge0.push_back(edge0);
    m_vCuttingPathEdge1.push_back(edge1);
    if (m_vCuttingPathEdge0.size() > maxNodes) m_vCuttingPathEdge0.erase(m_vCuttingPathEdge0.begin());
    if (m_vCuttingPathEdge1.size() > maxNodes) m_vCuttingPathEdge1.erase(m_vCuttingPathEdge1.begin());
    if (m_isSweptQuadValid) performCut(edge0, edge1, m_sweptQuad);
    return 1;
  }
  // Cutting::performCut(edge0, edge1, sweptQuad) (Cutting.cpp:537-566), with the edge pass the reference has commented out
  int performCut(const vec3d& edge0, const vec3d& edge1, vec3d sweptQuad[4]) {
    m_ctFacePoints = 0;
    m_ctEdgePoints = 0;
    computeFaceIntersections(edge0, edge1);
    computeEdgeIntersections(sweptQuad);
    return (int)(m_ctFacePoints + m_ctEdgePoints);
  }

  // -> number of face points, -1 without buffers (Cutting.cpp:169-251)
  int computeFaceIntersections(const vec3d& s0, const vec3d& s1) {
    if (!m_isMemBuffersLoaded) return -1;
    const double a[3] = {s0.x, s0.y, s0.z}, b[3] = {s1.x, s1.y, s1.z};
    int n = 0;
    check(fb_cut_face_intersections(h_, a, b, &n));
    m_ctFacePoints = (U32)n;
    return n;
  }
  // -> 1, -1 without buffers (Cutting.cpp:442-497); the count is kept in countEdgePoints()
  int computeEdgeIntersections(const vec3d sweptQuad[4]) {
    if (!m_isMemBuffersLoaded) return -1;
    double q[12];
    for (int i = 0; i < 4; i++) { q[3 * i] = sweptQuad[i].x; q[3 * i + 1] = sweptQuad[i].y; q[3 * i + 2] = sweptQuad[i].z; }
    int n = 0;
    check(fb_cut_edge_intersections(h_, q, &n));
    m_ctEdgePoints = (U32)n;
    return 1;
  }
  int computeFaceCentroids() {
    if (!m_isMemBuffersLoaded) return -1;
    check(fb_cut_face_centroids(h_));
    return 1;
  }
  // Cutting::computeFaceSegmentIntersectionTest (Cutting.cpp:324-440): the built-in known answer, (0, 0, 0, 1)
  int computeFaceSegmentIntersectionTest(float out[4]) const {
    const float tri[12] = {-1, 0, -1, 1, 1, 0, -1, 1, 0, 0, 1, 1}, s0[3] = {0, -1, 0}, s1[3] = {0, 1, 0};
    check(fb_cut_segment_triangles(m_device, 1, tri, s0, s1, out));
    return 1;
  }

  U32 countFacePoints() const { return m_ctFacePoints; }
  U32 countEdgePoints() const { return m_ctEdgePoints; }
  bool isSweptQuadValid() const { return m_isSweptQuadValid; }
  const vec3d* sweptQuad() const { return m_sweptQuad; }
  U32 countTets() const { return (U32)(m_vMeshTets.size() / 4); }
  // what m_lpDrawFacePoints / m_lpDrawEdgePoints hold: (x, y, z, 1) per face / edge, and the flags
  void readFacePoints(std::vector<U32>& flags, std::vector<float>& xyzw) const { read(FB_CUT_FACES, 4, flags, xyzw); }
  void readEdgePoints(std::vector<U32>& flags, std::vector<float>& xyzw) const { read(FB_CUT_EDGES, 6, flags, xyzw); }
  // the flagged faces / edges only, ascending ids
  void readHits(int what, std::vector<U32>& ids, std::vector<float>& xyzw) const {
    int n = 0;
    check(fb_cut_read_hits(h_, what, 0, nullptr, nullptr, &n));
    ids.resize((size_t)n); xyzw.resize(4 * (size_t)n);
    if (n) check(fb_cut_read_hits(h_, what, n, ids.data(), xyzw.data(), &n));
  }

 private:
  void init() {
    m_ctFacePoints = m_ctEdgePoints = 0;
    m_isMemBuffersLoaded = false;
    m_isSweptQuadValid = false;
  }
  bool upload() {
    cleanupMemBuffers();
    check(fb_cut_create(&h_, m_device, (int)(m_vMeshVertices.size() / 3), m_vMeshVertices.data(), (int)(m_vMeshTets.size() / 4), m_vMeshTets.data()));
    m_isMemBuffersLoaded = true;
    return true;
  }
  void read(int what, int per, std::vector<U32>& flags, std::vector<float>& xyzw) const {
    flags.resize((size_t)per * countTets()); xyzw.resize(4 * flags.size());
    check(fb_cut_read(h_, what, flags.data(), xyzw.data()));
  }
  static void check(int rc) {
    if (rc != FB_OK) throw std::runtime_error(fb_last_error());
  }

  std::vector<U32> m_vMeshTets;
  std::vector<double> m_vMeshVertices;
  U32 m_ctFacePoints, m_ctEdgePoints;
  Deformable* m_lpDeformable;
  bool m_isMemBuffersLoaded;
  bool m_isSweptQuadValid;
  vec3d m_sweptQuad[4];
  std::vector<vec3d> m_vCuttingPathEdge0, m_vCuttingPathEdge1;
  fb_cut_t h_;
  int m_device;
};

}  // namespace FEM
}  // namespace PS
