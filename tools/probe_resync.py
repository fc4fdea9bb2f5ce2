"""Development aid: where a re-sync (Deformable::syncForceModel) spends its time (FEMBRAIN_TIMING=1 prints the laps)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("FEMBRAIN_TIMING", "1")
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
g = FemIntegrator(v, t, fixed)
for k in range(3):
    t0 = time.perf_counter()
    g.resync(v, t, fixed)
    print("resync %d: %.1f ms" % (k, (time.perf_counter() - t0) * 1e3), flush=True)
