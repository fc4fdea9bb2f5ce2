"""Development aid: re-sync time when the mesh really changes between calls (a synthetic cut: the elements crossing a plane are
removed, each replaced by two elements on a new node at its centroid, nodes appended) -- buffer sizes change from call to call."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, synthetic_cut, truth_cube  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
axis = int(sys.argv[2]) if len(sys.argv) > 2 else 1
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
g = FemIntegrator(v, t, fixed)
meshes = [(v, t)]
for k in range(3):
    v2, t2, d = synthetic_cut(meshes[-1][0], meshes[-1][1], axis=axis, where=0.3 + 0.15 * k)
    meshes.append((v2, t2))
    print("cut %d: %d removed, %d added, %d new nodes" % (k, len(d["removed"]), len(d["added"]), len(d["new_xyz"])), flush=True)
for rep in range(2):
    for k, (vv, tt) in enumerate(meshes[1:] + meshes[:1]):
        t0 = time.perf_counter()
        g.resync(vv, tt, fixed)
        print("resync to mesh %d (%d nodes, %d tets): %.2f ms" % (k, len(vv), len(tt), (time.perf_counter() - t0) * 1e3), flush=True)
for k in range(3):
    t0 = time.perf_counter()
    g.resync(v, t, fixed)
    print("resync, same mesh: %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
