"""development aid: us per PCG iteration, FB_PCG_MERGED vs FB_PCG_PERSISTENT, on the bench meshes"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
for n in [int(a) for a in sys.argv[1:]] or [56]:
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    ref = None
    for name, var in (("merged", fl.FB_PCG_MERGED), ("persistent", fl.FB_PCG_PERSISTENT)):
        g = FemIntegrator(v, t, fixed, pcg_variant=var)
        res = []
        for k in range(4):
            g.reset_to_rest()
            g.set_uniform_force(1, -10000.0)
            it = g.do_timestep()
            res.append((it, g.last.solve_seconds / it * 1e6))
        q = g.get_q_state()[0]
        if ref is None:
            ref = q
        print("n=%d %-10s iterations %s us/iter %s  maxrel vs merged %.2e" % (n, name, [r[0] for r in res], ["%.2f" % r[1] for r in res],
              np.abs(q - ref).max() / np.abs(ref).max()), flush=True)
        g.close()
