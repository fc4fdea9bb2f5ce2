"""One step on the bench cube (56^3 nodes, 998,250 tets): the workload of tools/pmc_kernel.sh runs on the assembly kernels.
FEMBRAIN_ASM_KERNEL=rows|tets picks the kernel, FEMBRAIN_ASM_PROFILE=1 prints where k_assemble_tets' wavefronts spend their time."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube

n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
v, t = truth_cube(n, n, n, 0.1)
g = FemIntegrator(v, t, fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n)))
g.set_uniform_force(1, -10000.0)
print("iterations", g.do_timestep(), "assembly kernel", "tets" if g._L.fb_fem_assembly_kernel(g.h) else "rows")
