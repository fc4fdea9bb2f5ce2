"""Development aid: BASELINE config 2 sizes (27^3 cube = 105k tets, 2 slices per CU; 33^3 = 3; 37^3 = 4) -- us per PCG iteration of the
persistent solver with whole slices in LDS (k_pcg_pipe<..,5,16>, opt-in: FEMBRAIN_PIPE_SMALL=1) against the default (8, 8),
and, with FEMBRAIN_PERSIST_TIMING=1 in the environment, the phase table of each."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from fembrain_amd import lib as fl  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [27, 33, 37]:
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    for small in ("1", "0"):
        os.environ["FEMBRAIN_PIPE_SMALL"] = small
        g = FemIntegrator(v, t, fixed, matrix_precision=fl.FB_MATRIX_F32)
        its, sec = [], 0.0
        for k in range(4):
            g.reset_to_rest()
            g.set_uniform_force(1, -10000.0)
            it = g.do_timestep()
            if k:
                its.append(it)
                sec += g.last.solve_seconds
        p = g.pcg_path()
        print("n %d (%d tets, %d slices): %s  %s iterations  %.2f us per iteration (solve), persist_info %s" % (n, len(t), (len(v) + 63) // 64, p["kernel"], its, sec / sum(its) * 1e6, g.persist_info()), flush=True)
        g.close()
