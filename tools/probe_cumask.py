"""development aid: two processes on one GPU, each stepping a persistent handle on its own half of the CUs (FEMBRAIN_CU_MASK)"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1:
    half = int(sys.argv[1])
    os.environ["FEMBRAIN_CU_MASK"] = "%d:128" % (128 * half)
    import numpy as np
    from fembrain_amd import lib as fl
    from fembrain_amd.fem import FemIntegrator
    from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
    n = 40
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    g = FemIntegrator(v, t, fixed, pcg_variant=fl.FB_PCG_PERSISTENT)
    print(half, "info", g.persist_info(), flush=True)
    t0 = time.time()
    its = []
    for k in range(60):
        g.set_uniform_force(1, -10000.0)
        its.append(g.do_timestep())
    print(half, "steps done in %.2f s" % (time.time() - t0), its[:3], g.pcg_path(), "us/iter %.2f" % (g.last.solve_seconds / its[-1] * 1e6), flush=True)
else:
    ps = [subprocess.Popen([sys.executable, __file__, str(h)]) for h in (0, 1)]
    rc = [p.wait() for p in ps]
    print("rc", rc)
    sys.exit(max(rc))
