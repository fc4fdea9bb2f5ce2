"""Development aid: per-iteration and SpMV time with an alternative build of the library (argv: .so path, nodes per side)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd import lib as fl  # noqa: E402

fl.LIB_PATH = os.path.abspath(sys.argv[1])
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 56
v, t = truth_cube(n, n, n, 0.1)
g = FemIntegrator(v, t, fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n)))
its, solve = 0, 0.0
for k in range(3):
    g.set_uniform_force(1, -10000.0)
    it = g.do_timestep()
    if k:
        its += it
        solve += g.last.solve_seconds
sp = g.time_spmv(300)
print("%s n=%d: %.2f us/iteration (%d iterations), spmv %.2f us -> %.0f GB/s" %
      (os.path.basename(sys.argv[1]), n, solve / its * 1e6, its, sp * 1e6, g.spmv_bytes() / sp / 1e9), flush=True)
