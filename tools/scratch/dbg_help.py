import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from scipy.spatial import Delaunay
from fembrain_amd.meshgen import fixed_vertices_to_dofs
def mesh(m, seed=2):
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(float)
    pts = (g + rng.uniform(-0.35, 0.35, size=g.shape)) * 0.1
    t = Delaunay(pts).simplices.astype(np.int32)
    vol = np.einsum("ij,ij->i", pts[t[:, 1]] - pts[t[:, 0]], np.cross(pts[t[:, 2]] - pts[t[:, 0]], pts[t[:, 3]] - pts[t[:, 0]])) / 6
    keep = np.abs(vol) > 1e-9
    t, vol = t[keep], vol[keep]
    t[vol < 0] = t[vol < 0][:, [0, 2, 1, 3]]
    return pts, np.ascontiguousarray(t), fixed_vertices_to_dofs(np.nonzero(g[:, 0] == 0)[0])
v, t, fixed = mesh(16)
os.environ["FEMBRAIN_PERSIST_MIN_WAVES"] = "1"
os.environ["FEMBRAIN_PCG_PERSIST"] = "0"
gm = FemIntegrator(v, t, fixed, matrix_precision=fl.FB_MATRIX_F32)
del os.environ["FEMBRAIN_PCG_PERSIST"]
gm.set_uniform_force(1, -100.0)
_, rhs = gm.system()
itm, xm = gm.pcg(rhs, eps=1e-8, max_iter=20000)
x = np.random.default_rng(1).normal(size=len(rhs)); x[fixed] = 0
ym = gm.spmv(x)
print("two-launch", itm)
for mode, minlen, dbg in (("0", "16", None), ("1", "30", "1"), ("1", "30", "2"), ("1", "30", None), ("1", "16", "1"), ("1", "16", "2")):
    os.environ["FEMBRAIN_PIPE_HELPERS"] = mode
    os.environ["FEMBRAIN_PIPE_HELP_MINLEN"] = minlen
    os.environ.pop("FEMBRAIN_PIPE_HELP_DEBUG", None)
    if dbg:
        os.environ["FEMBRAIN_PIPE_HELP_DEBUG"] = dbg
    g = FemIntegrator(v, t, fixed, pcg_variant=fl.FB_PCG_PERSISTENT)
    g.set_uniform_force(1, -100.0)
    g.system()
    it, xp = g.pcg(rhs, eps=1e-8, max_iter=20000)
    print("helpers mode", mode, "minlen", minlen, "debug", dbg, g.pcg_path()["kernel"], "helper tasks", fl.lib().fb_fem_persist_helpers(g.h), "iterations", it, "dx", np.abs(xp - xm).max() / np.abs(xm).max())
    g.close()
print("---- a regular cube with helpers forced on short streams (well-conditioned: any logic error shows as a wrong count)")
from fembrain_amd.meshgen import cube_fixed_plane_i0, truth_cube
os.environ.pop("FEMBRAIN_PIPE_HELP_DEBUG", None)
for n in (14, 26):
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    res = []
    for mode in ("0", "1"):
        os.environ["FEMBRAIN_PIPE_HELPERS"] = mode
        os.environ["FEMBRAIN_PIPE_HELP_MINLEN"] = "4"
        g = FemIntegrator(v, t, fixed, pcg_variant=fl.FB_PCG_PERSISTENT)
        g.set_uniform_force(1, -10000.0)
        _, rhs = g.system()
        it, x = g.pcg(rhs, eps=1e-8, max_iter=20000)
        res.append((it, x))
        print("cube", n, "helpers", mode, g.pcg_path()["kernel"], "tasks", fl.lib().fb_fem_persist_helpers(g.h), "iterations", it)
        g.close()
    print("   dx", np.abs(res[0][1] - res[1][1]).max() / np.abs(res[0][1]).max())
