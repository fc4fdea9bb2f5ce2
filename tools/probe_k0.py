"""Development aid: the MFMA element-stiffness pass alone (for a PMC pass: MfmaUtil of k_element_k0)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
v, t = truth_cube(n, n, n, 0.1)
g = FemIntegrator(v, t, fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n)))
s = g.time_element_stiffness(5)
print("K0 pass: %.1f us, %.1f GFLOP/s fp64" % (s * 1e6, 2592.0 * len(t) / s / 1e9))
