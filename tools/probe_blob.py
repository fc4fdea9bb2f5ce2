"""Sweep + tetrahedralize a .blob model with 256 points on its longest axis (development aid)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd.blobtree import read_blob
from fembrain_amd.poly import GpuPoly
path = sys.argv[1]
b = read_blob(path)
lo, hi = b.bbox
cell = float((hi - lo).max()) / 254.0
p = GpuPoly(b)
dims = p.sweep(cell)
c = p.classify()
p.tetrahedralize()
s, t = p.time_pipeline(10)
npts = int(np.prod(dims))
print("%s: %d prims %d ops, grid %s cell %.5f: sweep %.1f us (%.0f Mvoxels/s), pipeline %.1f us (%.0f Mvoxels/s), %d tets" %
      (os.path.basename(path), b.n_prims, b.n_ops, dims, cell, s * 1e6, npts / s / 1e6, t * 1e6, npts / t / 1e6, c.n_included_cells * 6))
