"""development aid: PCG iterations and time of the first step from rest, Jacobi vs the opt-in 3x3 block-Jacobi"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
for n in [int(a) for a in sys.argv[1:]] or [27, 56]:
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    for name, var in (("jacobi (default)", fl.FB_PCG_MERGED), ("jacobi literal", fl.FB_PCG_REFERENCE), ("block-jacobi 3x3", fl.FB_PCG_BLOCK_JACOBI)):
        g = FemIntegrator(v, t, fixed, pcg_variant=var)
        out = []
        for k in range(3):
            g.set_uniform_force(1, -10000.0)
            it = g.do_timestep()
            out.append("%d its %.2f ms" % (it, g.last.solve_seconds * 1e3))
        print("n=%d %-18s %s" % (n, name, " | ".join(out)), flush=True)
        g.close()
