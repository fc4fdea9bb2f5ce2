// Host-only: reads .blob files with include/fembrain/BlobReader.h and prints the flat arrays; reads a .veg file and prints
// the mesh.  tests/test_cpp_host.py compares the output with the Python readers.  usage: read_models blob <file> | veg <file>
#include <cstdio>
#include <cstring>

#include "fembrain/BlobReader.h"

static void dump(const char* name, const std::vector<float>& v) {
  std::printf("%s %zu", name, v.size());
  for (size_t i = 0; i < v.size(); i++) std::printf(" %.9g", v[i]);
  std::printf("\n");
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  std::string err;
  if (!std::strcmp(argv[1], "blob")) {
    PS::SKETCH::LinearBlobTreeData d;
    if (!PS::SKETCH::readBlobFile(argv[2], d, &err)) { std::printf("ERROR %s\n", err.c_str()); return 1; }
    dump("header", d.header); dump("ops", d.ops); dump("prims", d.prims); dump("mtx", d.mtx); dump("pbox", d.primBoxes);
    return 0;
  }
  std::vector<double> v;
  std::vector<int> e;
  if (!PS::FEM::readVegFile(argv[2], v, e, &err)) { std::printf("ERROR %s\n", err.c_str()); return 1; }
  std::printf("vertices %zu", v.size());
  for (size_t i = 0; i < v.size(); i++) std::printf(" %.17g", v[i]);
  std::printf("\nelements %zu", e.size());
  for (size_t i = 0; i < e.size(); i++) std::printf(" %d", e[i]);
  std::printf("\n");
  return 0;
}
