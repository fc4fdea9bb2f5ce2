"""Development aid: the 256^3 sphere pipeline incl. the marching-cubes surface pass, for rocprofv3 --kernel-trace."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd.poly import GpuPoly, sphere_blob  # noqa: E402

p = GpuPoly(sphere_blob())
p.sweep_grid((-0.5, -0.5, -0.5), 1.0 / 254.0, (256, 256, 256))
p.classify()
p.tetrahedralize()
p.surface()
print("pipeline", p.time_pipeline(10), "surface", p.time_surface(10), p.counts.n_surface_vertices, flush=True)
