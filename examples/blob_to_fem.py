"""BlobTree model -> tet mesh -> deformable step, all on the device (BASELINE config 2 in miniature).

    python examples/blob_to_fem.py [model.blob] [cellsize]

Reads a FemBrain .blob model (default: the ventricle fixture), polygonizes it with the 6-tet rule, hands the tet mesh to the
FEM handle without a host copy, clamps the lowest tenth of the vertices and runs a few corotational steps under gravity."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd.blobtree import read_blob  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import fixed_vertices_to_dofs  # noqa: E402
from fembrain_amd.poly import GpuPoly  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "tests", "golden", "blob", "ventricle.blob")
cell = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
poly = GpuPoly(read_blob(path))
xyz, tets = poly.run_tetrahedralizer(cell)          # sweep + classification + tet emission
print("%s at cell %.3f: grid %s, %d vertices, %d tets" % (os.path.basename(path), cell, poly.dims, len(xyz), len(tets)))
low = np.nonzero(xyz[:, 1] <= np.percentile(xyz[:, 1], 10))[0]
fem = FemIntegrator.from_poly(poly, fixed_vertices_to_dofs(low))   # the mesh never leaves the device
for step in range(5):
    fem.set_uniform_force(1, -200.0)
    it = fem.do_timestep()
    q = fem.get_q_state()[0]
    print("step %d: %4d PCG iterations, assembly %.2f ms, solve %.2f ms, max |q| %.4f" %
          (step, it, fem.last.assembly_seconds * 1e3, fem.last.solve_seconds * 1e3, np.abs(q).max()))
verts, normals, tris = poly.run(cell)               # the render surface of the same model
print("surface: %d vertices, %d triangles" % (len(verts), len(tris)))
