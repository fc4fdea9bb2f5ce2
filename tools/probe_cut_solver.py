"""Development aid: the persistent solver on the truth cube after 0, 1 and 2 synthetic cuts (the mesh grows and gets rows of twice the
usual length along the cut): kernel, iterations, us per iteration, slice widths and SELL padding.  usage: probe_cut_solver.py [n=56]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from fembrain_amd import lib as fl  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, synthetic_cut, truth_cube  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
for cut in range(3):
    if cut:
        v, t, _ = synthetic_cut(v, t, axis=cut % 3, where=0.23 + 0.13 * (cut - 1))
    g = FemIntegrator(v, t, fixed)
    its, us = [], []
    for _ in range(3):
        g.reset_to_rest()
        g.set_uniform_force(1, -10000.0)
        its.append(g.do_timestep())
        us.append(round(g.last.solve_seconds / its[-1] * 1e6, 2))
    cnt = g._L.fb_fem_device_plan_get(g.h, b"slice_off", None, 0)
    so = np.zeros(cnt, np.int32)
    g._L.fb_fem_device_plan_get(g.h, b"slice_off", fl.iptr(so), cnt)
    w = np.diff(so)
    if os.environ.get("PROBE_DUMP"):
        np.save(os.path.join(os.environ["PROBE_DUMP"], "widths_cut%d.npy" % cut), w)
    # blocks of the pattern: node pairs sharing an element (+ the diagonal)
    a = np.concatenate([t[:, [i, j]] for i in range(4) for j in range(4)])
    nnz = len(np.unique(a[:, 0].astype(np.int64) * len(v) + a[:, 1]))
    print(json.dumps(dict(cuts=cut, nodes=len(v), tets=len(t), kernel=g.pcg_path()["kernel"], iterations=its, us_per_iteration=us, renumbering=g.renumbering(),
                          slices=int(len(w)), slots=int(so[-1]), blocks=int(nnz), padding=round(float(so[-1]) * 64 / nnz - 1, 3),
                          widths=dict(max=int(w.max()), mean=round(float(w.mean()), 2), p50=int(np.percentile(w, 50)), p90=int(np.percentile(w, 90)), p99=int(np.percentile(w, 99))),
                          spmv_mb=round(g.spmv_bytes() / 1e6, 1), persist=g.persist_info(), gather=g.persist_gather())), flush=True)
    g.close()
