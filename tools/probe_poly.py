"""Quick performance probe of the field path on the 256^3 sphere grid (development aid)."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd.poly import GpuPoly, sphere_blob
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = GpuPoly(sphere_blob())
dims = p.sweep_grid((-0.5, -0.5, -0.5), 1.0 / (n - 2), (n, n, n))
c = p.classify()
p.tetrahedralize()
s, t = p.time_pipeline(30)
npts = n ** 3
print("grid %d^3: sweep %.1f us (%.0f GB/s), pipeline %.1f us -> %.0f Mvoxels/s; tets %d verts %d" %
      (n, s * 1e6, npts * 16 / s / 1e9, t * 1e6, npts / t / 1e6, c.n_included_cells * 6, p.counts.n_tet_vertices))
st = p.time_stages(10)
print("stages us: sweep %.1f classify+scans %.1f vertices %.1f elements %.1f (all with events %.1f); sweep + xyzf grid %.1f us" % (st[0] * 1e6, st[1] * 1e6, st[2] * 1e6, st[3] * 1e6, st[4] * 1e6, p.time_grid(10) * 1e6))
