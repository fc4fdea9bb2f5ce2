// `.blob` and `.veg` readers for C++ hosts that do not link the reference's own ModelReader / VolMeshIO.  Header-only.
//
//   PS::SKETCH::readBlobFile   <-> ModelReader::read / readNode / readTransformation (reference
//                                  src/implicit/ReadSceneModel.cpp:238-750), PrepareAllBoxes (src/implicit/Polygonizer.cpp:210-600)
//                                  and LinearBlobTree::load (src/implicit/LinearBlobTree.cpp:43-167): flat INI file ->
//                                  the four flat arrays fb_poly_create takes (header 12, operator 16, primitive 20, matrix 12).
//   PS::FEM::readVegFile       <-> VolMeshIO::readVega (src/deformable/VolMeshIO.h:19-20): *VERTICES / *ELEMENTS TET, 1-indexed.
//
// Same logic as fembrain_amd/blobtree.py (the reader the parity tests use); tests/test_cpp_host.py compares the two on
// every fixture model.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "GPUPoly.h"

namespace PS {
namespace SKETCH {
namespace blobio {

typedef std::map<std::string, std::string> Section;

inline std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

inline bool parseIni(const char* path, std::map<std::string, Section>& out) {
  std::ifstream f(path);
  if (!f) return false;
  std::string line;
  Section* cur = nullptr;
  while (std::getline(f, line)) {
    line = trim(line);
    if (line.empty() || line[0] == ';' || line[0] == '#') continue;
    if (line[0] == '[' && line[line.size() - 1] == ']') {
      cur = &out[trim(line.substr(1, line.size() - 2))];
      continue;
    }
    const size_t eq = line.find('=');
    if (eq != std::string::npos && cur) (*cur)[trim(line.substr(0, eq))] = trim(line.substr(eq + 1));
  }
  return true;
}

inline std::vector<double> numbers(const std::string& s) {
  std::vector<double> v;
  const char* p = s.c_str();
  while (*p) {
    if ((*p >= '0' && *p <= '9') || ((*p == '-' || *p == '+' || *p == '.') && ((p[1] >= '0' && p[1] <= '9') || p[1] == '.'))) {
      char* end = nullptr;
      v.push_back(std::strtod(p, &end));
      if (end == p) break;
      p = end;
    } else {
      p++;
    }
  }
  return v;
}

inline bool has(const Section& s, const char* k) { return s.find(k) != s.end(); }
inline std::string str(const Section& s, const char* k, const char* d = "") { Section::const_iterator it = s.find(k); return it == s.end() ? std::string(d) : it->second; }
inline double num(const Section& s, const char* k, double d) { return has(s, k) ? std::atof(s.find(k)->second.c_str()) : d; }
inline bool flag(const Section& s, const char* k) { const std::string v = str(s, k, "0"); return v == "1" || v == "true" || v == "True" || v == "TRUE"; }
inline void vec(const Section& s, const char* k, int n, double dflt, double* out) {
  std::vector<double> v = has(s, k) ? numbers(s.find(k)->second) : std::vector<double>();
  for (int i = 0; i < n; i++) out[i] = i < (int)v.size() ? v[i] : dflt;
}

struct Mat4 {
  float m[16];
};
inline Mat4 identity() { Mat4 r; for (int i = 0; i < 16; i++) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f; return r; }
inline Mat4 mul(const Mat4& a, const Mat4& b) {
  Mat4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float s = 0.0f;
      for (int k = 0; k < 4; k++) s += a.m[4 * i + k] * b.m[4 * k + j];
      r.m[4 * i + j] = s;
    }
  return r;
}
// forward matrix T * R * S of a node (readTransformation, ReadSceneModel.cpp:693-750), float arithmetic
inline Mat4 affine(const Section& sec) {
  double s[3], q[4], t[3];
  vec(sec, "AffineScale", 3, 1.0, s);
  vec(sec, "AffineRotate", 4, 0.0, q);
  vec(sec, "AffineTranslate", 3, 0.0, t);
  if (!has(sec, "AffineRotate")) { q[0] = q[1] = q[2] = 0.0; q[3] = 1.0; }
  const float x = (float)q[0], y = (float)q[1], z = (float)q[2], w = (float)q[3];
  Mat4 R = identity(), T = identity(), S = identity();
  R.m[0] = 1 - 2 * (y * y + z * z); R.m[1] = 2 * (x * y - w * z); R.m[2] = 2 * (x * z + w * y);
  R.m[4] = 2 * (x * y + w * z); R.m[5] = 1 - 2 * (x * x + z * z); R.m[6] = 2 * (y * z - w * x);
  R.m[8] = 2 * (x * z - w * y); R.m[9] = 2 * (y * z + w * x); R.m[10] = 1 - 2 * (x * x + y * y);
  T.m[3] = (float)t[0]; T.m[7] = (float)t[1]; T.m[11] = (float)t[2];
  S.m[0] = (float)s[0]; S.m[5] = (float)s[1]; S.m[10] = (float)s[2];
  return mul(mul(T, R), S);
}
inline bool isIdentity(const Mat4& a) { const Mat4 i = identity(); return std::memcmp(a.m, i.m, sizeof a.m) == 0; }
// inverse of an affine 4x4 in double, rows 0..2 returned as 12 floats
inline bool inverseRows(const Mat4& a, float out[12]) {
  const double A[9] = {a.m[0], a.m[1], a.m[2], a.m[4], a.m[5], a.m[6], a.m[8], a.m[9], a.m[10]}, t[3] = {a.m[3], a.m[7], a.m[11]};
  const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
  if (det == 0.0 || det != det) return false;
  double I[9];
  I[0] = (A[4] * A[8] - A[5] * A[7]) / det; I[1] = (A[2] * A[7] - A[1] * A[8]) / det; I[2] = (A[1] * A[5] - A[2] * A[4]) / det;
  I[3] = (A[5] * A[6] - A[3] * A[8]) / det; I[4] = (A[0] * A[8] - A[2] * A[6]) / det; I[5] = (A[2] * A[3] - A[0] * A[5]) / det;
  I[6] = (A[3] * A[7] - A[4] * A[6]) / det; I[7] = (A[1] * A[6] - A[0] * A[7]) / det; I[8] = (A[0] * A[4] - A[1] * A[3]) / det;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) out[4 * r + c] = (float)I[3 * r + c];
    out[4 * r + 3] = (float)-(I[3 * r] * t[0] + I[3 * r + 1] * t[1] + I[3 * r + 2] * t[2]);
  }
  return true;
}

struct Box {
  float lo[3], hi[3];
};
inline Box mapBox(const Mat4& M, const Box& b) {  // lo and hi corner only, as the reference does (mat.mapAffine of the two)
  Box r;
  for (int a = 0; a < 3; a++) {
    const float u = M.m[4 * a] * b.lo[0] + M.m[4 * a + 1] * b.lo[1] + M.m[4 * a + 2] * b.lo[2] + M.m[4 * a + 3];
    const float v = M.m[4 * a] * b.hi[0] + M.m[4 * a + 1] * b.hi[1] + M.m[4 * a + 2] * b.hi[2] + M.m[4 * a + 3];
    r.lo[a] = std::min(u, v); r.hi[a] = std::max(u, v);
  }
  return r;
}

struct Prim {
  int type, im;
  float pos[3], dir[3], res[3], color[3];
};
struct Op {
  int type, flags, lc, rc;
  float res[4];
  Box box;
  bool has_box;
};

enum { ofRightOp = 1, ofLeftOp = 2, ofRange = 4, ofUnary = 8, ofIsRight = 16, ofBreak = 32 };

inline int primType(const std::string& n) {
  static const char* names[] = {"POINT", "LINE", "CYLINDER", "DISC", "RING", "CUBE", "TRIANGLE", "QUADRICPOINT", "NULL", "INSTANCE"};
  for (int i = 0; i < 10; i++) if (n == names[i]) return i;
  return -1;
}
inline int opType(const std::string& n) {
  static const char* names[] = {"UNION", "INTERSECTION", "DIFFERENCE", "SMOOTH DIFFERENCE", "BLEND", "RICCI BLEND", "", "FASTQUADRICPOINTSET",
                                "CACHE", "TWIST", "TAPER", "BEND", "SHEAR"};
  for (int i = 0; i < 13; i++) if (names[i][0] && n == names[i]) return i;
  return 0;
}

// PrepareAllPrimBBoxes (Polygonizer.cpp:268-420): skeleton grown by ISO_VALUE = 0.5
inline Box primBox(const Prim& p) {
  const float off = 0.5f;
  Box b;
  for (int a = 0; a < 3; a++) { b.lo[a] = FLT_MAX; b.hi[a] = FLT_MIN; }
  switch (p.type) {
    case 0: case 8: for (int a = 0; a < 3; a++) { b.lo[a] = p.pos[a] - off; b.hi[a] = p.pos[a] + off; } break;
    case 1: for (int a = 0; a < 3; a++) { const float e = off + 3.0f * off * (p.dir[a] - p.pos[a]); b.lo[a] = p.pos[a] - e; b.hi[a] = p.dir[a] + e; } break;
    case 3: case 4: {
      const float radius = p.res[0] + off;
      for (int a = 0; a < 3; a++) { const float e = (radius + off) * (1.0f - p.dir[a]) + off * p.dir[a]; b.lo[a] = p.pos[a] - e; b.hi[a] = p.pos[a] + e; }
    } break;
    case 2: for (int a = 0; a < 3; a++) {
      const float s1 = p.pos[a] + p.res[1] * p.dir[a], e = (off + p.res[0]) + 0.5f * off * p.dir[a];
      b.lo[a] = p.pos[a] - e; b.hi[a] = s1 + e;
    } break;
    case 5: for (int a = 0; a < 3; a++) { const float side = p.res[0] + off; b.lo[a] = p.pos[a] - side; b.hi[a] = p.pos[a] + side; } break;
    case 6: for (int a = 0; a < 3; a++) {
      b.lo[a] = std::min(std::min(p.pos[a], p.dir[a]), p.res[a]) - off; b.hi[a] = std::max(std::max(p.pos[a], p.dir[a]), p.res[a]) + off;
    } break;
    case 7: for (int a = 0; a < 3; a++) { const float w = p.dir[1] + off; b.lo[a] = p.pos[a] - w; b.hi[a] = p.pos[a] + w; } break;
    case 9: for (int a = 0; a < 3; a++) b.lo[a] = b.hi[a] = 0.0f; break;
    default: break;
  }
  return b;
}

struct Reader {
  std::map<std::string, Section> ini;
  std::vector<Prim> prims;
  std::vector<Op> ops;
  std::vector<Mat4> fwd;          // box matrices (forward), index 0 = identity
  std::vector<float> inv;         // 12 per matrix node
  std::map<int, int> script2array;
  std::string err;

  bool fail(const std::string& m) { if (err.empty()) err = m; return false; }

  // returns the array index; *isOp tells which array
  bool readNode(int nid, int* index, bool* isOp) {
    char name[64];
    std::snprintf(name, sizeof name, "BLOBNODE %d", nid);
    std::map<std::string, Section>::const_iterator it = ini.find(name);
    if (it == ini.end()) return fail(std::string("missing [") + name + "]");
    const Section& sec = it->second;
    if (flag(sec, "IsOperator")) {
      const int idx = (int)ops.size();
      if (!script2array.count(nid)) script2array[nid] = idx;
      Op op;
      std::memset(&op, 0, sizeof op);
      op.type = opType(str(sec, "OperatorType"));
      ops.push_back(op);
      if (op.type == 5) {
        const float power = (float)num(sec, "power", 1.0);
        ops[idx].res[0] = power; ops[idx].res[1] = 1.0f / power;
      } else if (op.type >= 9 && op.type <= 12) {
        ops[idx].flags |= ofUnary;
        ops[idx].res[0] = (float)num(sec, has(sec, "factor") ? "factor" : "rate", 1.0);
      }
      if (flag(sec, "ChildrenIDsUseRange")) {
        std::vector<double> rng = numbers(str(sec, "ChildrenIDsRange"));
        if (rng.size() < 2) return fail("bad ChildrenIDsRange");
        int first = -1, last = -1;
        for (int c = (int)rng[0]; c <= (int)rng[1]; c++) {
          int ci; bool cop;
          if (!readNode(c, &ci, &cop)) return false;
          if (first < 0) first = ci;
          last = ci;
        }
        ops[idx].lc = first; ops[idx].rc = last; ops[idx].flags |= ofRange;
      } else {
        std::vector<double> ids = numbers(str(sec, "ChildrenIDs"));
        const bool binary = !(ops[idx].flags & ofUnary);
        if (ids.empty() || (binary && ids.size() != 2)) return fail("operator node with a wrong number of children");
        int lc; bool lop;
        if (!readNode((int)ids[0], &lc, &lop)) return false;
        ops[idx].lc = lc;
        if (lop) ops[idx].flags |= ofLeftOp;
        if (binary) {
          int rc; bool rop;
          if (!readNode((int)ids[1], &rc, &rop)) return false;
          ops[idx].rc = rc;
          if (rop) { ops[idx].flags |= ofRightOp; ops[rc].flags |= ofIsRight; }
          if (lop && rop) { ops[lc].flags |= ofBreak; ops[rc].flags |= ofBreak; }
        }
      }
      *index = idx; *isOp = true;
      return true;
    }
    const int idx = (int)prims.size();
    Prim p;
    std::memset(&p, 0, sizeof p);
    p.type = primType(str(sec, "PrimitiveType"));
    if (p.type < 0) return fail("unknown primitive type '" + str(sec, "PrimitiveType") + "'");
    double a[3], b[3], c[3];
    switch (p.type) {
      case 7: {
        vec(sec, "position", 3, 0.0, a);
        const float scale = (float)num(sec, "scale", 0.0), radius = (float)num(sec, "radius", 0.0);
        for (int k = 0; k < 3; k++) p.pos[k] = (float)a[k];
        p.dir[0] = scale; p.dir[1] = radius; p.dir[2] = radius * radius;
        p.res[0] = scale / (radius * radius * radius * radius); p.res[1] = (-2.0f * scale) / (radius * radius); p.res[2] = scale;
      } break;
      case 0: vec(sec, "position", 3, 0.0, a); for (int k = 0; k < 3; k++) p.pos[k] = (float)a[k]; break;
      case 1: vec(sec, "start", 3, 0.0, a); vec(sec, "end", 3, 0.0, b); for (int k = 0; k < 3; k++) { p.pos[k] = (float)a[k]; p.dir[k] = (float)b[k]; } break;
      case 3: case 4:
        vec(sec, "position", 3, 0.0, a); vec(sec, "direction", 3, 0.0, b);
        for (int k = 0; k < 3; k++) { p.pos[k] = (float)a[k]; p.dir[k] = (float)b[k]; }
        p.res[0] = (float)num(sec, "radius", 0.0);
        break;
      case 2:
        vec(sec, "position", 3, 0.0, a); vec(sec, "direction", 3, 0.0, b);
        for (int k = 0; k < 3; k++) { p.pos[k] = (float)a[k]; p.dir[k] = (float)b[k]; }
        p.res[0] = (float)num(sec, "radius", 0.0); p.res[1] = (float)num(sec, "height", 0.0);
        break;
      case 5: vec(sec, "position", 3, 0.0, a); for (int k = 0; k < 3; k++) p.pos[k] = (float)a[k]; p.res[0] = (float)num(sec, "side", 0.0); break;
      case 6:
        vec(sec, "corner0", 3, 0.0, a); vec(sec, "corner1", 3, 0.0, b); vec(sec, "corner2", 3, 0.0, c);
        for (int k = 0; k < 3; k++) { p.pos[k] = (float)a[k]; p.dir[k] = (float)b[k]; p.res[k] = (float)c[k]; }
        break;
      case 9: {
        const int isop = (int)num(sec, "OriginalNodeIsOp", 0.0);
        p.res[0] = 0.0f; p.res[1] = (float)num(sec, "OriginalNodeIndex", 0.0); p.res[2] = (float)isop;
        const std::string on = str(sec, "OriginalNodeType");
        p.dir[0] = (float)(isop ? opType(on) : std::max(0, primType(on)));
      } break;
      default: break;
    }
    double col[4];
    vec(sec, "MtrlDiffused", 4, 0.0, col);
    for (int k = 0; k < 3; k++) p.color[k] = (float)col[k];
    const Mat4 F = affine(sec);
    p.im = 0;
    if (!isIdentity(F)) {
      float rows[12];
      if (!inverseRows(F, rows)) return fail("singular node transformation");
      p.im = (int)fwd.size();
      fwd.push_back(F);
      inv.insert(inv.end(), rows, rows + 12);
    }
    prims.push_back(p);
    if (!script2array.count(nid)) script2array[nid] = idx;
    *index = idx; *isOp = false;
    return true;
  }

  Box opBox(int i, const std::vector<Box>& pb) {
    Op& op = ops[i];
    Box b;
    if (op.flags & ofRange) {
      b = pb[op.lc];
      for (int k = op.lc + 1; k <= op.rc; k++)
        for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], pb[k].lo[a]); b.hi[a] = std::max(b.hi[a], pb[k].hi[a]); }
    } else {
      b = (op.flags & ofLeftOp) ? opBox(op.lc, pb) : pb[op.lc];
      if (!(op.flags & ofUnary)) {
        const Box r = (op.flags & ofRightOp) ? opBox(op.rc, pb) : pb[op.rc];
        for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], r.lo[a]); b.hi[a] = std::max(b.hi[a], r.hi[a]); }
      }
    }
    op.box = b; op.has_box = true;
    return b;
  }
};

}  // namespace blobio

// Reads a .blob model into the flat arrays of LinearBlobTreeData; false + *error on a malformed file.
inline bool readBlobFile(const char* path, LinearBlobTreeData& out, std::string* error = nullptr) {
  using namespace blobio;
  Reader R;
  auto bad = [&](const std::string& m) { if (error) *error = std::string(path) + ": " + m; return false; };
  if (!parseIni(path, R.ini)) return bad("cannot open");
  const Section& g = R.ini["Global"];
  if ((int)num(g, "FileVersion", 0.0) < 1) return bad("invalid file version");
  std::vector<double> roots = numbers(str(g, "RootIDs"));
  if (roots.empty()) return bad("no RootIDs");
  R.fwd.push_back(identity());
  const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  R.inv.assign(ident, ident + 12);
  int idx; bool isop;
  if (!R.readNode((int)roots[0], &idx, &isop)) return bad(R.err);
  if (R.prims.empty()) return bad("no primitives");
  for (size_t i = 0; i < R.prims.size(); i++) {  // setAllInstancedNodes (ReadSceneModel.cpp:214-236)
    Prim& p = R.prims[i];
    if (p.type != 9) continue;
    const int sid = (int)p.res[1];
    if (!R.script2array.count(sid)) return bad("instance of an unknown node");
    const int origin = R.script2array[sid];
    if (origin < 0 || origin >= (int)(p.res[2] != 0.0f ? R.ops.size() : R.prims.size())) return bad("instance: original index out of range");
    p.res[0] = (float)origin;
  }
  // PrepareAllBoxes (Polygonizer.cpp:210-263): primitives, operators, instanced nodes, operators again, model box
  std::vector<Box> pb(R.prims.size());
  for (size_t i = 0; i < R.prims.size(); i++) {
    pb[i] = primBox(R.prims[i]);
    if (R.prims[i].im != 0) pb[i] = mapBox(R.fwd[R.prims[i].im], pb[i]);
  }
  if (!R.ops.empty()) R.opBox(0, pb);
  bool any_instance = false;
  for (size_t i = 0; i < R.prims.size(); i++) any_instance = any_instance || R.prims[i].type == 9;
  if (any_instance) {
    for (size_t i = 0; i < R.prims.size(); i++) {
      const Prim& p = R.prims[i];
      if (p.type != 9) continue;
      const int origin = (int)p.res[0];
      const Box src = p.res[2] != 0.0f ? R.ops[origin].box : pb[origin];
      pb[i] = mapBox(R.fwd[p.im], src);
    }
    if (!R.ops.empty()) R.opBox(0, pb);
  }
  float mlo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mhi[3] = {FLT_MIN, FLT_MIN, FLT_MIN};
  for (size_t i = 0; i < pb.size(); i++)
    for (int a = 0; a < 3; a++) { mlo[a] = std::min(mlo[a], pb[i].lo[a]); mhi[a] = std::max(mhi[a], pb[i].hi[a]); }
  const float NULL_BLOB = 65535.0f;
  out.header.assign(12, 0.0f);
  for (int a = 0; a < 3; a++) { out.header[a] = mlo[a]; out.header[4 + a] = mhi[a]; }
  out.header[3] = out.header[7] = 1.0f;
  out.header[8] = (float)R.prims.size(); out.header[9] = (float)R.ops.size(); out.header[10] = (float)R.fwd.size(); out.header[11] = NULL_BLOB;
  out.prims.assign(20 * R.prims.size(), 0.0f);
  for (size_t i = 0; i < R.prims.size(); i++) {
    float* P = &out.prims[20 * i];
    const Prim& p = R.prims[i];
    P[0] = (float)p.type; P[1] = (float)p.im;
    for (int k = 0; k < 3; k++) { P[4 + k] = p.pos[k]; P[8 + k] = p.dir[k]; P[12 + k] = p.res[k]; P[16 + k] = p.color[k]; }
    P[19] = 1.0f;
  }
  out.ops.assign(16 * R.ops.size(), 0.0f);
  for (size_t i = 0; i < R.ops.size(); i++) {
    float* O = &out.ops[16 * i];
    const Op& op = R.ops[i];
    O[0] = (float)op.type; O[1] = (float)op.lc; O[2] = (float)op.rc; O[3] = NULL_BLOB;
    O[4] = op.res[0]; O[5] = op.res[1]; O[6] = op.res[2]; O[7] = (float)op.flags;
    if (op.has_box) for (int a = 0; a < 3; a++) { O[8 + a] = op.box.lo[a]; O[12 + a] = op.box.hi[a]; }
    O[11] = O[15] = 1.0f;
  }
  out.mtx = R.inv;
  out.primBoxes.assign(6 * pb.size(), 0.0f);
  for (size_t i = 0; i < pb.size(); i++)
    for (int a = 0; a < 3; a++) { out.primBoxes[6 * i + a] = pb[i].lo[a]; out.primBoxes[6 * i + 3 + a] = pb[i].hi[a]; }
  return true;
}

}  // namespace SKETCH

namespace FEM {

// Vega text mesh (data/models/blobtree/peanut.veg): "*VERTICES\n n 3 0 0\n id x y z ..." and "*ELEMENTS\nTET\n m 4 0\n id a b c d ..." with
// 1-indexed ids; materials and regions are ignored on this path.  elements come back 0-indexed.
inline bool readVegFile(const char* path, std::vector<double>& vertices, std::vector<int>& elements, std::string* error = nullptr) {
  std::ifstream f(path);
  if (!f) { if (error) *error = std::string(path) + ": cannot open"; return false; }
  vertices.clear(); elements.clear();
  std::string line;
  int mode = 0;  // 1 vertices header, 2 vertices, 3 "TET", 4 elements header, 5 elements
  size_t nv = 0, ne = 0;
  while (std::getline(f, line)) {
    line = SKETCH::blobio::trim(line);
    if (line.empty() || line[0] == '#') continue;
    if (line[0] == '*') {
      mode = line.compare(0, 9, "*VERTICES") == 0 ? 1 : (line.compare(0, 9, "*ELEMENTS") == 0 ? 3 : 0);
      continue;
    }
    std::istringstream is(line);
    if (mode == 1) { is >> nv; vertices.reserve(3 * nv); mode = 2; }
    else if (mode == 2) { long id; double x, y, z; if (is >> id >> x >> y >> z) { vertices.push_back(x); vertices.push_back(y); vertices.push_back(z); } }
    else if (mode == 3) { if (line.compare(0, 3, "TET") != 0) { if (error) *error = std::string(path) + ": only TET elements are supported"; return false; } mode = 4; }
    else if (mode == 4) { is >> ne; elements.reserve(4 * ne); mode = 5; }
    else if (mode == 5) { long id, a, b, c, d; if (is >> id >> a >> b >> c >> d) { elements.push_back((int)a - 1); elements.push_back((int)b - 1); elements.push_back((int)c - 1); elements.push_back((int)d - 1); } }
  }
  if (vertices.size() != 3 * nv || elements.size() != 4 * ne || nv == 0) {
    if (error) *error = std::string(path) + ": vertex / element counts do not match the headers";
    return false;
  }
  return true;
}

}  // namespace FEM
}  // namespace PS
