"""Development aid: the polygonizer's own mesh (ventricle.blob at cell sizes giving ~100k / ~1M tets) through the FEM handle: kernel, us per
PCG iteration, how the gathered vector is laid out and the line counts behind it.  usage: probe_blob_gather.py [cellsize ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from fembrain_amd.blobtree import read_blob  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import fixed_vertices_to_dofs  # noqa: E402
from fembrain_amd.poly import GpuPoly  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for cs in [float(a) for a in sys.argv[1:]] or [0.115, 0.055]:
    poly = GpuPoly(read_blob(os.path.join(ROOT, "tests", "golden", "blob", "ventricle.blob")))
    xyz, tets = poly.run_tetrahedralizer(cs)
    v = xyz.astype(np.float64)
    ycut = np.sort(v[:, 1])[len(v) // 20]
    g = FemIntegrator.from_poly(poly, fixed_vertices_to_dofs(np.nonzero(v[:, 1] <= ycut)[0]))
    its, us = [], []
    for _ in range(3):
        g.reset_to_rest()
        g.set_uniform_force(1, -10000.0)
        its.append(g.do_timestep())
        us.append(round(g.last.solve_seconds / its[-1] * 1e6, 2))
    print(dict(cellsize=cs, nodes=len(v), tets=len(tets), kernel=g.pcg_path()["kernel"], iterations=its, us_per_iteration=us, persist=g.persist_info(),
               gather=g.persist_gather(), renumbering=g.renumbering(), helpers=int(g._L.fb_fem_persist_helpers(g.h))), flush=True)
    g.close()
