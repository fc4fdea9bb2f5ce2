import os, sys
sys.path.insert(0, '/root/repo')
import subprocess
code = r'''
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
n = int(sys.argv[1])
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
g = FemIntegrator(v, t, fixed)
out = []
for k in range(3):
    g.reset_to_rest()
    g.set_uniform_force(1, -10000.0)
    it = g.do_timestep()
    out.append("%d its %.2f us/it" % (it, g.last.solve_seconds / it * 1e6))
print("n=%d mask=%s kernel=%s persist=%s | %s" % (n, os.environ.get("FEMBRAIN_CU_MASK", "-"), g.pcg_path()["kernel"], g.persist_info(), " | ".join(out)), flush=True)
'''
open('/tmp/_fb_cm.py','w').write(code)
for n in (27, 20, 36):
    for mask in (None, "0:32", "0:64", "0:96", "0:128", "0:192"):
        env = dict(os.environ)
        if mask: env["FEMBRAIN_CU_MASK"] = mask
        subprocess.run([sys.executable, '/tmp/_fb_cm.py', str(n)], env=env)
