"""Development aid: repeated create / step / destroy of both handle kinds, watching device memory and results."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402  (only for mem_get_info)
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402
from fembrain_amd.cutting import FB_CUT_EDGES, FB_CUT_FACES, Cutting  # noqa: E402
from fembrain_amd.poly import GpuPoly, sphere_blob  # noqa: E402

v, t = truth_cube(16, 16, 16, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(16, 16))
free0 = None
ref = None
for cycle in range(60):
    g = FemIntegrator(v, t, fixed)
    for _ in range(3):
        g.set_uniform_force(1, -10000.0)
        g.do_timestep()
    q = g.get_q_state()[0]
    if cycle % 7 == 3:
        g.resync(v, t, fixed)
    g.close()
    p = GpuPoly(sphere_blob())
    p.sweep_grid((-0.5, -0.5, -0.5), 1.0 / 62.0, (64, 64, 64))
    p.classify(); p.tetrahedralize(); p.surface()
    n = p.counts.n_tets
    col = p.read_surface_colors()
    p.close()
    s = GpuPoly(sphere_blob())
    s.sweep_slab((-0.5, -0.5, -0.5), 1.0 / 62.0, (64, 64, 64), 15, 20)
    s.classify(); s.tetrahedralize()
    sx, st = s.read_tetmesh_slab(16, 16, 16, 1000)
    s.close()
    c = Cutting(v, t)
    nf = c.compute_face_intersections((0.013, -1.0, 0.021), (0.013, 3.0, 0.021))
    ne = c.compute_edge_intersections([(-9, 0.73, -8.4), (9.2, 0.73, -9.1), (-8.9, 0.73, 9.3), (9.05, 0.73, 8.8)])
    ids, _ = c.read_hits(FB_CUT_EDGES)
    c.close()
    n = (n, len(col), len(sx), len(st), nf, ne, int(ids.sum()))
    if ref is None:
        ref = (q.copy(), n)
    assert np.array_equal(q, ref[0]) and n == ref[1], "results changed between cycles"
    free = torch.cuda.mem_get_info()[0]
    if cycle == 5:
        free0 = free
    if cycle > 5 and cycle % 10 == 0:
        print("cycle %d free %.1f MiB (delta %.2f MiB)" % (cycle, free / 2 ** 20, (free - free0) / 2 ** 20), flush=True)
# long run of one handle: 400 steps, graph replay path, values stay finite and settle
g = FemIntegrator(v, t, fixed)
its = []
for k in range(400):
    g.set_uniform_force(1, -10000.0)
    its.append(g.do_timestep())
q = g.get_q_state()[0]
print("400 steps: iterations first/last %d/%d, max|q| %.4f, finite %s" % (its[0], its[-1], np.abs(q).max(), np.isfinite(q).all()))
