"""Assembly time at the bench mesh with both kernels (FEMBRAIN_ASM_KERNEL).  usage: python tools/probe_assembly.py [n=56]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube

n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
for kern in ("tets", "tets1", "rows"):
    os.environ["FEMBRAIN_ASM_KERNEL"] = kern
    g = FemIntegrator(v, t, fixed)
    g.set_uniform_force(1, -10000.0)
    g.do_timestep()
    print(kern, "kernel", fl.lib().fb_fem_assembly_kernel(g.h), "tets", len(t), "assembly %.1f us" % (g.time_assembly(20) * 1e6), flush=True)
    g.close()
