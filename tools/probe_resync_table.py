"""Development aid: fb_fem_resync_delta against fb_fem_resync over mesh size and size of the change (elements crossing a plane removed and
appended again, mirrored: the sizes stay constant, so the figures are free of re-allocations).  Writes a JSON table."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from fembrain_amd import lib as fl  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import apply_delta, cube_fixed_plane_i0, fixed_vertices_to_dofs, synthetic_cut, truth_cube  # noqa: E402

rows = []
for n in [int(a) for a in sys.argv[2:]] or [27, 56, 70]:
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    for cut_first in (False, True):
        g = FemIntegrator(v, t, fixed)
        ref = FemIntegrator(v, t, fixed)
        cv, ct = v, t
        if cut_first:   # (a mesh that has been cut: appended nodes, an internal node order, irregular rows)
            cv, ct, d = synthetic_cut(cv, ct, axis=1, where=0.23)
            g.resync_delta(d, fixed, track=False)
            ref.resync(cv, ct, fixed)
        for frac in (0.001, 0.01, 0.05):
            want = max(1, int(len(ct) * frac))
            dts, drs = [], []
            for k in range(4):
                x = cv[:, 1][ct]
                lo, hi = cv[:, 1].min(), cv[:, 1].max()
                c = lo + (0.35 + 0.06 * k) * (hi - lo)
                hit = np.nonzero((x.min(axis=1) < c + 0.3 * frac ** 0.3) & (x.max(axis=1) > c))[0].astype(np.int32)
                hit = np.sort(hit[:: max(1, len(hit) // want)][:want])
                d = dict(removed=hit, changed_ids=np.zeros(0, np.int32), changed_nodes=np.zeros(0, np.int32), added=ct[hit][:, [1, 0, 2, 3]], new_xyz=np.zeros(0))
                v2, t2 = apply_delta(cv, ct, d)
                t0 = time.perf_counter()
                g.resync_delta(d, fixed, track=False)
                dts.append((time.perf_counter() - t0) * 1e3)
                t0 = time.perf_counter()
                ref.resync(v2, t2, fixed)
                drs.append((time.perf_counter() - t0) * 1e3)
                assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED
                cv, ct = v2, t2
            row = dict(nodes=len(cv), tets=len(ct), mesh="cut once" if cut_first else "grid cube", renumbered=bool(g.renumbering()[0]), elements_changed=len(hit),
                       fraction=round(len(hit) / len(ct), 5), delta_ms=round(min(dts[1:]), 3), full_ms=round(min(drs[1:]), 3))
            print(json.dumps(row), flush=True)
            rows.append(row)
        g.close()
        ref.close()
json.dump(rows, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/resync_table.json", "w"), indent=1)
