"""development aid: PCG iterations and time of the first steps from rest, Jacobi vs the opt-in 3x3 block-Jacobi (two-launch and inside
the persistent kernel)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
for n in [int(a) for a in sys.argv[1:]] or [27, 56]:
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    for name, var, env in (("jacobi (default)", fl.FB_PCG_MERGED, None), ("jacobi literal", fl.FB_PCG_REFERENCE, None),
                           ("block-jacobi 3x3, two-launch", fl.FB_PCG_BLOCK_JACOBI, "0"), ("block-jacobi 3x3, persistent", fl.FB_PCG_BLOCK_JACOBI, None)):
        if env is not None:
            os.environ["FEMBRAIN_PCG_PERSIST"] = env
        g = FemIntegrator(v, t, fixed, pcg_variant=var)
        os.environ.pop("FEMBRAIN_PCG_PERSIST", None)
        out = []
        for k in range(4):
            g.set_uniform_force(1, -10000.0)
            it = g.do_timestep()
            out.append("%d its %.2f ms (%.2f us/it)" % (it, g.last.solve_seconds * 1e3, g.last.solve_seconds * 1e6 / max(it, 1)))
        print("n=%d %-30s %-32s %s" % (n, name, g.pcg_path()["kernel"] or "-", " | ".join(out)), flush=True)
        g.close()
