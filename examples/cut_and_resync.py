"""A blade through a cantilever: intersection passes of the cutting tool, then a topology change and the re-sync.

    python examples/cut_and_resync.py [nodes per side]

The subdivision of the cut tets is the host application's business (CuttableMesh::cut in FemBrain); here the tets the blade
touches are simply removed, which is enough to show the device passes and Deformable::syncForceModel -> fb_fem_resync."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd.cutting import FB_CUT_EDGES, Cutting  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
fem = FemIntegrator(v, t, fixed)
fem.set_uniform_force(1, -2000.0)
fem.do_timestep()
cur = v + fem.get_q_state()[0].reshape(-1, 3)
cut = Cutting(cur, t)                                  # current (deformed) positions, as Cutting::createMemBuffers takes them
x = 0.1 * (n // 4)
quad = [(x + 0.013, -1, -9), (x + 0.013, -1, 9), (x + 0.013, 0.1 * n * 0.4, -9), (x + 0.013, 0.1 * n * 0.4, 9)]
hits = cut.compute_edge_intersections(quad)
ids, pts = cut.read_hits(FB_CUT_EDGES)
touched = np.unique(ids // 6)
print("%d tets: blade cuts %d edges of %d tets" % (len(t), hits, len(touched)))
keep = np.ones(len(t), bool)
keep[touched] = False
t0 = time.perf_counter()
fem.resync(v, t[keep], fixed)                          # plan rebuilt on the device
print("re-sync with %d tets: %.1f ms" % (keep.sum(), (time.perf_counter() - t0) * 1e3))
fem.set_uniform_force(1, -2000.0)
print("next step: %d PCG iterations" % fem.do_timestep())
