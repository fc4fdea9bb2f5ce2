"""Development aid: one large mesh on one GPU (argv: nodes per side) -- create, 2 steps, memory in use."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402  (mem_get_info only)
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
t0 = time.perf_counter()
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
t1 = time.perf_counter()
free0 = torch.cuda.mem_get_info()[0]
g = FemIntegrator(v, t, fixed)
t2 = time.perf_counter()
print("n=%d: %d nodes, %d tets; mesh %.1f s, create %.2f s, device memory %.1f GiB" %
      (n, len(v), len(t), t1 - t0, t2 - t1, (free0 - torch.cuda.mem_get_info()[0]) / 2 ** 30), flush=True)
for k in range(2):
    g.set_uniform_force(1, -10000.0)
    ts = time.perf_counter()
    it = g.do_timestep()
    print("step %d: %.2f s, %d iterations, %.1f us/iteration" % (k, time.perf_counter() - ts, it, g.last.solve_seconds / it * 1e6), flush=True)
sp = g.time_spmv(20)
print("spmv %.1f us -> %.0f GB/s" % (sp * 1e6, g.spmv_bytes() / sp / 1e9))
