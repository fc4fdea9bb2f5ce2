import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
n = 56
v0, t0 = truth_cube(n, n, n, 0.1)
m = np.random.default_rng(12345).permutation(len(v0))
v = np.empty_like(v0); v[m] = v0
t = np.ascontiguousarray(m[t0].astype(np.int32))
fixed = fixed_vertices_to_dofs(np.sort(m[cube_fixed_plane_i0(n, n)]))
g = FemIntegrator(v, t, fixed)
for k in range(3):
    g.resync(v, t, fixed)
os.environ["FEMBRAIN_TIMING"] = "1"
g2 = FemIntegrator(v0, t0, fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n)))
