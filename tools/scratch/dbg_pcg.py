import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator, bsr_to_scipy
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, synthetic_cut, truth_cube
n = 9
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
v2, t2, d = synthetic_cut(v, t, axis=1, where=0.4)
v3, t3, d2 = synthetic_cut(v2, t2, axis=2, where=0.55, every_changed=2, stride=6)
f = np.zeros(3 * len(v3)); f[1::3] = -10000.0
for prec in (fl.FB_MATRIX_F32, fl.FB_MATRIX_F64):
    for variant in (fl.FB_PCG_MERGED, fl.FB_PCG_REFERENCE):
        g = FemIntegrator(v3, t3, fixed, renumber=fl.FB_RENUMBER_OFF, matrix_precision=prec, pcg_variant=variant)
        g.set_external_forces(f); it0 = g.do_timestep()
        g.set_external_forces(f); it1 = g.do_timestep()
        K, rhs = g.system()
        bptr, bcol = g.pattern()
        A = bsr_to_scipy(bptr, bcol, K)
        dg = A.diagonal()
        its, x = g.pcg(rhs, 1e-6, 10000)
        r = rhs - A @ x
        print("prec", prec, "variant", variant, "steps", it0, it1, "pcg(rhs) its", its, "true rel resid (M^-1 norm)", np.sqrt((r * r / dg).sum() / (rhs * rhs / dg).sum()))
        for eps in (1e-7, 1e-8, 1e-9):
            its2, x2 = g.pcg(rhs, eps, 10000)
            r2 = rhs - A @ x2
            print("    eps", eps, its2, np.sqrt((r2 * r2 / dg).sum() / (rhs * rhs / dg).sum()), "dx vs eps1e-6", np.abs(x - x2).max() / np.abs(x2).max())
        g.close()
print("sensitivity: fp32 merged / fp64 reference with the first step's tolerance nudged")
for prec, variant in ((fl.FB_MATRIX_F32, fl.FB_PCG_MERGED), (fl.FB_MATRIX_F64, fl.FB_PCG_REFERENCE), (fl.FB_MATRIX_F64, fl.FB_PCG_MERGED)):
    for eps1 in (1e-6, 0.97e-6, 0.9e-6, 0.5e-6, 1e-7, 1e-9):
        g = FemIntegrator(v3, t3, fixed, renumber=fl.FB_RENUMBER_OFF, matrix_precision=prec, pcg_variant=variant)
        g.set_cg(eps1, 10000)
        g.set_external_forces(f); it0 = g.do_timestep()
        g.set_cg(1e-6, 10000)
        g.set_external_forces(f); it1 = g.do_timestep()
        print(prec, variant, eps1, it0, it1)
        g.close()
