// C++ adaptor with the reference's polygonizer surface over the C ABI (fembrain_hip.h).  Header-only; link with
// -lfembrain_hip.
//
//   PS::SKETCH::GPUPoly        <-> class GPUPoly : SG::SGMesh (reference src/implicit/OclPolygonizer.h:45-231): the
//                                  field / classification / tetrahedralizer members; the GL buffer members of the
//                                  original stay in the host application.
//   PS::SKETCH::FieldComputer  <-> reference src/implicit/FieldComputer.h:31-70 (voxel-grid sweep, point fields)
//   PS::SKETCH::LinearBlobTreeData: the four flat arrays of LinearBlobTree (src/implicit/LinearBlobTree.h:20-82)
#pragma once
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../fembrain_hip.h"

namespace PS {
namespace SKETCH {

typedef unsigned int U32;
typedef unsigned char U8;

struct LinearBlobTreeData {
  std::vector<float> header;  // 12
  std::vector<float> ops;     // 16 per operator
  std::vector<float> prims;   // 20 per primitive
  std::vector<float> mtx;     // 12 per matrix node (index 0 = identity)
  std::vector<float> primBoxes;  // optional, 6 per primitive (lo, hi): PrepareAllBoxes' primitive boxes (BlobReader.h fills them);
                                 // only FB_FIELD_CPU_BOX reads them
  int countPrimitives() const { return (int)(prims.size() / 20); }
  int countOperators() const { return (int)(ops.size() / 16); }
  int countMtxNodes() const { return (int)(mtx.size() / 12); }
};

struct vec4u { U32 x, y, z, w; };

class GPUPoly {
 public:
  GPUPoly() : h_(nullptr), m_cellsize(0.14f), m_device(0) {}  // DEFAULT_CELL_SIZE, src/implicit/Polygonizer.h
  explicit GPUPoly(const LinearBlobTreeData& blob, int device = 0) : h_(nullptr), m_cellsize(0.14f), m_device(device) { setBlob(blob); }
  ~GPUPoly() { fb_poly_destroy(h_); }
  GPUPoly(const GPUPoly&) = delete;
  GPUPoly& operator=(const GPUPoly&) = delete;

  // deep copy of the tree (OclPolygonizer.cpp:1602-1648)
  bool setBlob(const LinearBlobTreeData& blob) {
    fb_poly_destroy(h_);
    h_ = nullptr;
    m_primBoxes = blob.primBoxes.size() == 6 * (size_t)blob.countPrimitives() ? blob.primBoxes : std::vector<float>();
    return fb_poly_create(&h_, m_device, blob.header.data(), blob.countOperators(), blob.ops.data(), blob.countPrimitives(), blob.prims.data(),
                          blob.countMtxNodes(), blob.mtx.data()) == FB_OK;
  }
  // Which of the reference's two field evaluations to follow (fembrain_hip.h: fb_poly_set_field_semantics): FB_FIELD_CPU
  // (default), FB_FIELD_CPU_BOX (needs blob.primBoxes at setBlob time) or FB_FIELD_OPENCL -- what the OpenCL GPUPoly this
  // class replaces computes, operator-index defect included.
  bool setFieldSemantics(int semantics) {
    return fb_poly_set_field_semantics(h_, semantics, m_primBoxes.empty() ? nullptr : m_primBoxes.data()) == FB_OK;
  }
  int fieldSemantics() const { return fb_poly_field_semantics(h_); }
  void setCellSize(float c) { m_cellsize = c; }
  float cellsize() const { return m_cellsize; }

  // field sweep, classification and the marching-cubes surface; 1 ok, -1 if cellsize < 0.01 or nothing crosses the
  // surface (OclPolygonizer.cpp:644-757)
  int run() {
    int dims[3];
    if (fb_poly_sweep(h_, m_cellsize, dims) != FB_OK) return -1;
    if (fb_poly_classify(h_, &m_counts) != FB_OK) return -1;
    if (m_counts.n_crossed_edges == 0) return -1;
    return fb_poly_surface(h_, &m_counts) == FB_OK ? 1 : -1;
  }
  // GPUPoly::readbackMeshV3T3 (OclPolygonizer.cpp:1696-1744)
  bool readbackMeshV3T3(U32& ctVertices, std::vector<float>& vertices, U32& ctTriangles, std::vector<U32>& elements) const {
    ctVertices = (U32)m_counts.n_surface_vertices; ctTriangles = (U32)m_counts.n_surface_indices / 3;
    vertices.resize(3 * (size_t)ctVertices); elements.resize(3 * (size_t)ctTriangles);
    return fb_poly_read_surface(h_, vertices.data(), nullptr, elements.data()) == FB_OK;
  }
  // GPUPoly::readBackNormals
  bool readBackNormals(U32& ctVertices, std::vector<float>& verticesXYZ, std::vector<float>& normals) const {
    ctVertices = (U32)m_counts.n_surface_vertices;
    verticesXYZ.resize(3 * (size_t)ctVertices); normals.resize(3 * (size_t)ctVertices);
    return fb_poly_read_surface(h_, verticesXYZ.data(), normals.data(), nullptr) == FB_OK;
  }
  // the colour buffer of the surface mesh (m_outputMesh's vertex colours, written by ComputeVertexAttribs): RGBA per vertex
  bool readBackColors(U32& ctVertices, std::vector<float>& colorsRGBA) const {
    ctVertices = (U32)m_counts.n_surface_vertices;
    colorsRGBA.resize(4 * (size_t)ctVertices);
    return fb_poly_read_surface_colors(h_, colorsRGBA.data()) == FB_OK;
  }
  // FieldComputer::fieldValueAndColor for an array of points
  bool computeFieldAndColorArray(U32 ctVertices, float* xyzf, float* rgb) const { return fb_poly_field_color_array(h_, (int)ctVertices, xyzf, rgb) == FB_OK; }
  // GPUPoly::computeOffSurfacePointsAndFields (OclPolygonizer.cpp:1045-1107); `interval` is unused there as well
  int computeOffSurfacePointsAndFields(U32 /*interval*/, float len, U32& ctOutVertices, std::vector<float>& outOffSurfacePoints) {
    ctOutVertices = 2 * (U32)m_counts.n_surface_vertices;
    outOffSurfacePoints.resize(4 * (size_t)ctOutVertices);
    return fb_poly_off_surface(h_, len, outOffSurfacePoints.data()) == FB_OK ? 1 : -1;
  }
  // GPUPoly::applyFemDisplacements (OclPolygonizer.cpp:1543-1596) on the surface mesh; `deformed` receives what the
  // reference writes into its vertex VBO
  bool applyFemDisplacements(U32 dof, const double* displacements, std::vector<float>* deformed = nullptr, int mesh = FB_MESH_SURFACE) {
    if (deformed) deformed->resize(dof);
    return fb_poly_apply_displacements(h_, mesh, (int)dof, displacements, deformed ? deformed->data() : nullptr) == FB_OK;
  }
  // OclPolygonizer.cpp:762-819: needs run() first; returns the number of tets or -1
  int runTetrahedralizer() {
    if (fb_poly_tetrahedralize(h_, &m_counts) != FB_OK) return -1;
    return m_counts.n_tets;
  }
  bool readbackTetMesh(U32& ctVertices, std::vector<float>& vertices, U32& ctTets, std::vector<U32>& elements) const {
    ctVertices = (U32)m_counts.n_tet_vertices; ctTets = (U32)m_counts.n_tets;
    vertices.resize(3 * (size_t)ctVertices); elements.resize(4 * (size_t)ctTets);
    return fb_poly_read_tetmesh(h_, vertices.data(), elements.data()) == FB_OK;
  }
  // Multi-GPU field path (one GPUPoly per rank; fembrain_hip.h "z-slabs of one grid"): this rank's z-slab of the grid
  // `dims` through the tetrahedralizer.  allGatherVertexCounts(mine) must return the owned-vertex counts of ALL ranks in
  // rank order (one all-gather of an int: RCCL, MPI, ...).  The pieces of all ranks in rank order are the one-GPU mesh.
  template <typename AllGather>
  bool runTetrahedralizerSlab(const float lower[3], const int dims[3], int rank, int world, AllGather allGatherVertexCounts, U32& ctVertices,
                              std::vector<float>& vertices, U32& ctTets, std::vector<U32>& elements) {
    if (world < 1 || rank < 0 || rank >= world || dims[2] < 2 * world) return false;
    const int planes = dims[2], p0 = (int)((long long)planes * rank / world), p1 = (int)((long long)planes * (rank + 1) / world);
    const int zFirst = p0 > 0 ? p0 - 1 : 0, zLast = p1 + 1 < planes - 1 ? p1 + 1 : planes - 1;
    const int ownPlanes = p1 - p0, ownLayers = (p1 < planes - 1 ? p1 : planes - 1) - p0;
    if (fb_poly_sweep_slab(h_, lower, m_cellsize, dims, zFirst, zLast - zFirst + 1) != FB_OK) return false;
    if (fb_poly_classify(h_, &m_counts) != FB_OK || fb_poly_tetrahedralize(h_, &m_counts) != FB_OK) return false;
    int nv = 0, nt = 0;
    if (fb_poly_slab_counts(h_, p0, ownPlanes, ownLayers, &nv, &nt) != FB_OK) return false;
    const std::vector<int> all = allGatherVertexCounts(nv);
    unsigned int base = 0;
    for (int r = 0; r < rank && r < (int)all.size(); r++) base += (unsigned int)all[r];
    ctVertices = (U32)nv; ctTets = (U32)nt;
    vertices.resize(3 * (size_t)nv); elements.resize(4 * (size_t)nt);
    return fb_poly_read_tetmesh_slab(h_, p0, ownPlanes, ownLayers, base, vertices.data(), elements.data()) == FB_OK;
  }
  // GPUPoly::readBackVoxelGridSamples (OclPolygonizer.cpp:916-940)
  bool readBackVoxelGridSamples(vec4u& dim, std::vector<float>& arrXYZF) const {
    dim.x = (U32)m_counts.grid[0]; dim.y = (U32)m_counts.grid[1]; dim.z = (U32)m_counts.grid[2]; dim.w = (U32)m_counts.n_points;
    arrXYZF.resize(4 * (size_t)m_counts.n_points);
    return fb_poly_read_grid(h_, arrXYZF.data()) == FB_OK;
  }
  // GPUPoly::computeFieldArray (OclPolygonizer.cpp:943-986): step must be 4 (x,y,z,f), .w overwritten
  int computeFieldArray(U32 ctVertices, U32 step, std::vector<float>& vertices) const {
    if (step != 4 || vertices.size() < 4 * (size_t)ctVertices) return -1;
    return fb_poly_field_array(h_, (int)ctVertices, vertices.data()) == FB_OK ? 1 : -1;
  }
  // render-loop coupling when the FEM mesh is this grid's own tet mesh: surface vertices follow the displacements of
  // the two tet-mesh nodes of their grid edge (fb_poly_interpolate_displacements)
  bool applyTetMeshDisplacements(U32 tetDof, const double* displacements, std::vector<float>* deformed = nullptr) {
    if (deformed) deformed->resize(3 * (size_t)m_counts.n_surface_vertices);
    return fb_poly_interpolate_displacements(h_, (int)tetDof, displacements, deformed ? deformed->data() : nullptr) == FB_OK;
  }
  vec4u voxelGridDim() const { vec4u d = {(U32)m_counts.grid[0], (U32)m_counts.grid[1], (U32)m_counts.grid[2], (U32)m_counts.n_points}; return d; }
  U32 countSurfaceVoxels() const { return (U32)m_counts.n_surface_cells; }
  // GPUPoly::surfaceVoxels (filled by run(), OclPolygonizer.cpp:698-721): lower corners (x, y, z) of the cells the surface
  // crosses, in the reference's order (x outermost, z innermost); `lower` = model box lower corner (header[0..2])
  std::vector<float> surfaceVoxels(const float lower[3]) const {
    std::vector<float> out;
    std::vector<U8> cfg((size_t)m_counts.n_cells);
    if (cfg.empty() || fb_poly_read_classification(h_, nullptr, nullptr, cfg.data()) != FB_OK) return out;
    const int cx = m_counts.grid[0] - 1, cy = m_counts.grid[1] - 1, cz = m_counts.grid[2] - 1;
    out.reserve(3 * (size_t)m_counts.n_surface_cells);
    for (int i = 0; i < cx; i++)
      for (int j = 0; j < cy; j++)
        for (int k = 0; k < cz; k++) {
          const U8 c = cfg[(size_t)k * cx * cy + (size_t)j * cx + i];
          if (c != 0 && c != 255) { out.push_back(lower[0] + i * m_cellsize); out.push_back(lower[1] + j * m_cellsize); out.push_back(lower[2] + k * m_cellsize); }
        }
    return out;
  }
  const fb_poly_counts& counts() const { return m_counts; }

  // GPUPoly::storeTetMeshInVegaFormat (OclPolygonizer.cpp:1651-1694)
  bool storeTetMeshInVegaFormat(const char* path) const {
    U32 nv, nt;
    std::vector<float> v;
    std::vector<U32> e;
    if (!readbackTetMesh(nv, v, nt, e)) return false;
    FILE* f = std::fopen(path, "w");
    if (!f) return false;
    std::fprintf(f, "# Vega Mesh File, Generated by FemBrain.\n# %u vertices, %u elements\n\n*VERTICES\n%u 3 0 0\n", nv, nt, nv);
    for (U32 i = 0; i < nv; i++) std::fprintf(f, "%u %g %g %g\n", i + 1, v[3 * i], v[3 * i + 1], v[3 * i + 2]);
    std::fprintf(f, "\n*ELEMENTS\nTET\n%u 4 0\n", nt);
    for (U32 i = 0; i < nt; i++) std::fprintf(f, "%u %u %u %u %u\n", i + 1, e[4 * i] + 1, e[4 * i + 1] + 1, e[4 * i + 2] + 1, e[4 * i + 3] + 1);
    std::fprintf(f, "\n*MATERIAL BODY\nENU, 1000, 10000000, 0.45\n\n*REGION\nallElements, BODY\n");
    std::fclose(f);
    return true;
  }

  // GPUPoly::computeDevice() returned the OpenCL wrapper; there is none any more ("must become opaque/nullable")
  void* computeDevice() const { return nullptr; }
  // the C handle, e.g. for PS::FEM::HipIntegrator(fb_poly_t, ...): the tet mesh goes to the FEM on the device
  fb_poly_t handle() const { return h_; }

 private:
  fb_poly_t h_;
  float m_cellsize;
  std::vector<float> m_primBoxes;
  int m_device;
  fb_poly_counts m_counts;
};

class FieldComputer {
 public:
  explicit FieldComputer(const LinearBlobTreeData& blob, int device = 0) : h_(nullptr) {
    if (fb_poly_create(&h_, device, blob.header.data(), blob.countOperators(), blob.ops.data(), blob.countPrimitives(), blob.prims.data(),
                       blob.countMtxNodes(), blob.mtx.data()) != FB_OK)
      throw std::runtime_error(std::string("fembrain_hip: ") + fb_last_error());
  }
  ~FieldComputer() { fb_poly_destroy(h_); }
  FieldComputer(const FieldComputer&) = delete;
  FieldComputer& operator=(const FieldComputer&) = delete;
  // FieldComputer::fieldsForVoxelGrid (FieldComputer.cpp:143-231): returns the number of grid points or a negative code
  int fieldsForVoxelGrid(float cellsize, bool /*stackless*/ = true) {
    int dims[3];
    if (fb_poly_sweep(h_, cellsize, dims) != FB_OK) return -1;
    return dims[0] * dims[1] * dims[2];
  }
  float field(float x, float y, float z) {
    float p[4] = {x, y, z, 0.0f};
    fb_poly_field_array(h_, 1, p);
    return p[3];
  }
  int field(U32 ctVertices, U32 step, std::vector<float>& vertices) {
    if (step != 4) return -1;
    return fb_poly_field_array(h_, (int)ctVertices, vertices.data()) == FB_OK ? 1 : -1;
  }
  fb_poly_t handle() const { return h_; }

 private:
  fb_poly_t h_;
};

}  // namespace SKETCH
}  // namespace PS
