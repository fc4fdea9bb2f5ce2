import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, synthetic_cut, truth_cube
from oracle.pyoracle import OrcFem
n = 9
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
v2, t2, d = synthetic_cut(v, t, axis=1, where=0.4)
v3, t3, d2 = synthetic_cut(v2, t2, axis=2, where=0.55, every_changed=2, stride=6)
o = OrcFem(v3, t3)
o.integrator(fixed)
f = np.zeros(o.r); f[1::3] = -10000.0
ios = []
for k in range(2):
    o.set_external_forces(f); ios.append(abs(o.step()))
print("oracle", ios)
for prec in (fl.FB_MATRIX_F32, fl.FB_MATRIX_F64, fl.FB_MATRIX_AUTO):
    for mode in ("delta", "fresh"):
        if mode == "delta":
            g = FemIntegrator(v, t, fixed, renumber=fl.FB_RENUMBER_OFF, matrix_precision=prec)
            g.resync_delta(d, fixed); g.resync_delta(d2, fixed)
        else:
            g = FemIntegrator(v3, t3, fixed, renumber=fl.FB_RENUMBER_OFF, matrix_precision=prec)
        its = []
        for k in range(2):
            g.set_external_forces(f); its.append(g.do_timestep())
        print(prec, mode, its, g.matrix_precision(), np.abs(g.get_q_state()[0] - o.get_state()[0]).max() / np.abs(o.get_state()[0]).max())
        g.close()
