"""Development aid: the persistent pipelined PCG on a truth cube (argv: nodes per side) solving its first right-hand side with iteration
caps 1, 2, 5, 29, 30, 31, 100 and without one, then one step; FEMBRAIN_PERSIST_TIMING=1 (or 2: also by wavefront, by XCD, slowest
workgroups) prints the phase clocks of the development build of k_pcg_pipe after every solve."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["FEMBRAIN_PERSIST_MIN_WAVES"] = "1"
import numpy as np
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
n = int(sys.argv[1])
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
print("n", n, "nodes", len(v), "slices", (len(v) + 63) // 64, flush=True)
g = FemIntegrator(v, t, fixed, pcg_variant=fl.FB_PCG_PERSISTENT)
print("created", g.persist_info(), g.pcg_path(), flush=True)
g.set_uniform_force(1, -10000.0)
_, rhs = g.system()
print("system ok", flush=True)
for mi in (1, 2, 5, 29, 30, 31, 100, 20000):
    it, x = g.pcg(rhs, eps=1e-6, max_iter=mi)
    print("pcg max_iter", mi, "->", it, float(np.abs(x).max()), flush=True)
print("step", g.do_timestep(), flush=True)
