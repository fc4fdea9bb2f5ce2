"""Development aid: steps/s of the PCG variants and SpMV kernels on a truth cube (argv: nodes per side)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd import lib as fl  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
for variant, vname in ((fl.FB_PCG_MERGED, "merged"), (fl.FB_PCG_REFERENCE, "reference"), (fl.FB_PCG_PERSISTENT, "persistent")):
    for spmv, sname in ((fl.FB_SPMV_ROWS, "rows"), (fl.FB_SPMV_SPLIT, "split")):
        try:
            g = FemIntegrator(v, t, fixed, pcg_variant=variant, spmv_kernel=spmv)
        except fl.FbError as e:
            print(vname, sname, "n/a:", e)
            continue
        its, solve = [], 0.0
        for k in range(6):
            g.set_uniform_force(1, -10000.0)
            it = g.do_timestep()
            if k >= 2:
                its.append(it)
                solve += g.last.solve_seconds
        print("%-9s %-5s %7.2f us/iteration  (%.0f iterations/step, %.1f steps/s solve only)" %
              (vname, sname, solve / sum(its) * 1e6, np.mean(its), len(its) / solve))
        g.close()
