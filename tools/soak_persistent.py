"""Development aid: the persistent solver under long runs -- hundreds of steps on cubes of 3..24 slices per CU and on an unstructured
Delaunay mesh (every workgroup polls all flags), handles created / re-synced / destroyed in between, two handles alternating on one
device; no persistent launch may time out (fallbacks == 0), states stay finite, a repeated run gives the same bits."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd import lib as fl  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402


def cube(n):
    v, t = truth_cube(n, n, n, 0.1)
    return v, t, fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))


def delaunay(npts):
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(5)
    v = rng.uniform(0, 1, size=(npts, 3))
    t = Delaunay(v).simplices.astype(np.int32)
    vol = np.einsum("ij,ij->i", v[t[:, 1]] - v[t[:, 0]], np.cross(v[t[:, 2]] - v[t[:, 0]], v[t[:, 3]] - v[t[:, 0]])) / 6
    t = np.ascontiguousarray(t[np.abs(vol) > 1e-9])
    return v, t, fixed_vertices_to_dofs(np.nonzero(v[:, 0] < 0.05)[0])


def run(name, v, t, fixed, steps, load):
    t0 = time.time()
    out = []
    for rep in range(2):
        g = FemIntegrator(v, t, fixed)
        its = []
        for k in range(steps):
            g.set_uniform_force(1, load * (1.0 if k % 7 else -0.5))
            its.append(g.do_timestep())
            if k == steps // 2:
                g.resync(v, t, fixed)   # a re-sync in the middle: plan, producer lists and flags are rebuilt
        q = g.get_q_state()[0]
        p = g.pcg_path()
        assert p["fallbacks"] == 0 and np.isfinite(q).all(), (name, p)
        out.append((its, q, p))
        g.close()
    same = out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    print("%-12s %s: %d steps x 2, iterations %d..%d, launches %d, max producers %d, repeat bitwise %s, %.1f s" % (
        name, out[0][2]["kernel"], steps, min(out[0][0]), max(out[0][0]), out[0][2]["launches"], out[0][2]["max_producers"], same, time.time() - t0), flush=True)
    assert same, name


for n, steps in ((34, 150), (40, 150), (52, 80), (56, 80), (60, 40), (73, 20)):
    run("cube%d" % n, *cube(n), steps, -10000.0)
run("delaunay60k", *delaunay(60000), 60, -50.0)
# round 5: the table-driven instantiation with helpers and the node-by-node vector (606k-tet jittered lattice), the two-row kernel with the
# node-by-node vector (1.44M tets), the cube after a cut (slices dealt by slots, idle wavefronts helping at 12 slices per CU)
from fembrain_amd.meshgen import delaunay_jittered, synthetic_cut  # noqa: E402
for m, steps in ((45, 30), (60, 10)):
    dv, dt_, dfv = delaunay_jittered(m)
    run("jittered%d" % m, dv, dt_, fixed_vertices_to_dofs(dfv), steps, -100.0)
cv, ct, cf = cube(56)
cv, ct, _ = synthetic_cut(cv, ct, axis=1, where=0.23)
run("cube56_cut", cv, ct, cf, 30, -10000.0)
# two handles alternating on one device: their persistent launches serialise on their streams' turns, none may starve the other
a, b = FemIntegrator(*cube(40)), FemIntegrator(*cube(44))
for k in range(100):
    for h in (a, b):
        h.set_uniform_force(1, -10000.0)
        h.do_timestep()
assert a.pcg_path()["fallbacks"] == 0 and b.pcg_path()["fallbacks"] == 0
print("two handles alternating: 100 steps each, no fallback", flush=True)
