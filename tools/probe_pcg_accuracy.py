"""True residuals of the PCG variants on the same systems: the truth cube n^3 after two loaded steps, its own right-hand side and
`k` perturbed ones; for every solve the iteration count, |b - A x| / |b| (A x by fb_fem_spmv in fp64 on the stored matrix) and the
distance of x from the two-launch solver's x.  The persistent pipelined solver and the two-launch iteration both stop on a RECURRENCE
residual; this shows what the true one is when they do.   python tools/probe_pcg_accuracy.py [n=56] [k=12] [eps=1e-6]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fembrain_amd import lib as fl  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
os.environ["FEMBRAIN_PCG_PERSIST"] = "0"
g2 = FemIntegrator(v, t, fixed)
del os.environ["FEMBRAIN_PCG_PERSIST"]
gp = FemIntegrator(v, t, fixed)
print("persistent:", gp.pcg_path()["kernel"], "| two-launch:", g2.pcg_path()["kernel"] or "(none)")
f = np.zeros(gp.r)
f[1::3] = -10000.0
f[0::3] = 300.0 * np.sin(np.arange(len(v)))
for _ in range(2):
    g2.set_external_forces(f)
    g2.do_timestep()
gp.set_q_state(*g2.get_q_state())     # the SAME state on both handles: the same matrix bits
for g in (g2, gp):
    g.set_external_forces(f)
_, rhs2 = g2.system()
_, rhsp = gp.system()
print("rhs of the third step, persistent vs two-launch state: %.2e" % (np.abs(rhs2 - rhsp).max() / np.abs(rhs2).max()))
rng = np.random.default_rng(5)
worst = {"persistent": 0.0, "two-launch": 0.0}
for j in range(k + 1):
    b = rhs2 * (1.0 + (0.3 * rng.standard_normal(len(rhs2)) if j else 0.0))
    b[fixed] = 0.0
    out = {}
    for name, g in (("two-launch", g2), ("persistent", gp)):
        it, x = g.pcg(b, eps)
        res = np.linalg.norm(b - g.spmv(x)) / np.linalg.norm(b)
        out[name] = (it, res, x)
        worst[name] = max(worst[name], res)
    dx = np.abs(out["persistent"][2] - out["two-launch"][2]).max() / np.abs(out["two-launch"][2]).max()
    print("rhs %2d: two-launch %5d its |b-Ax|/|b| %.2e   persistent %5d its %.2e   max|dx|/max|x| %.2e" % (
        j, out["two-launch"][0], out["two-launch"][1], out["persistent"][0], out["persistent"][1], dx), flush=True)
print("worst true residual: two-launch %.2e, persistent %.2e (eps %g)" % (worst["two-launch"], worst["persistent"], eps))
