"""Development aid: fb_fem_resync_delta against fb_fem_resync at the n^3 cube.  First real cuts (meshgen.synthetic_cut: the mesh grows),
then changes of CONSTANT size on the cut mesh -- a slab of elements removed and appended again, mirrored -- which time the re-sync
itself, free of re-allocations.  FEMBRAIN_TIMING=1 prints the laps."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from fembrain_amd import lib as fl  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import apply_delta, cube_fixed_plane_i0, fixed_vertices_to_dofs, synthetic_cut, truth_cube  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
axis = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cuts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
names = {fl.FB_RESYNC_FULL: "full", fl.FB_RESYNC_DELTA_MERGED: "list updated", fl.FB_RESYNC_DELTA_REBUILT: "rebuilt on the device"}
def run():
    global g, ref, cv, ct
    g = FemIntegrator(v, t, fixed, expect_cuts=os.environ.get("PROBE_EXPECT_CUTS", "1") != "0")   # (the caller says it will cut: fb_fem_params.expect_cuts)
    ref = FemIntegrator(v, t, fixed)
    cv, ct = v, t


    def both(d, v2, t2, what):
        t0 = time.perf_counter()
        g.resync_delta(d, fixed, track=False)
        dt = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        ref.resync(v2, t2, fixed)
        dr = (time.perf_counter() - t0) * 1e3
        its = []
        for h in (g, ref):
            h.set_uniform_force(1, -10000.0)
            its.append(h.do_timestep())
        print("%s: -%d ~%d +%d elements, +%d nodes -> %d tets | delta %.2f ms (%s) | full %.2f ms | order %s / %s | %s / %s | %d / %d iterations, %.1f / %.1f us each"
              % (what, len(d["removed"]), len(d["changed_ids"]), len(d["added"]), len(d["new_xyz"]), len(t2), dt, names[g.resync_path()], dr, g.renumbering(), ref.renumbering(),
                 g.pcg_path()["kernel"], ref.pcg_path()["kernel"], its[0], its[1], g.last.solve_seconds / its[0] * 1e6, ref.last.solve_seconds / its[1] * 1e6), flush=True)


    for k in range(cuts):
        v2, t2, d = synthetic_cut(cv, ct, axis=(axis + k) % 3, where=0.23 + 0.13 * k)
        both(d, v2, t2, "cut %d" % k)
        cv, ct = v2, t2
    rng = np.random.default_rng(1)
    for k in range(6):
        # the elements crossing a plane (about 1 % of the mesh at 56^3), removed and appended again with two nodes swapped
        x = cv[:, axis][ct]
        c = cv[:, axis].min() + (0.31 + 0.07 * k) * (cv[:, axis].max() - cv[:, axis].min())
        hit = np.nonzero((x.min(axis=1) < c) & (x.max(axis=1) > c))[0].astype(np.int32)
        hit = hit[:: max(1, len(hit) // (len(ct) // 100))][: len(ct) // 100]
        d = dict(removed=hit, changed_ids=np.zeros(0, np.int32), changed_nodes=np.zeros(0, np.int32), added=ct[hit][:, [1, 0, 2, 3]], new_xyz=np.zeros(0))
        v2, t2 = apply_delta(cv, ct, d)
        both(d, v2, t2, "1 %% of the elements re-appended (%d)" % k)
        cv, ct = v2, t2
    g.close()
    ref.close()


for rep in range(int(os.environ.get("PROBE_REPS", "1"))):
    print("---- pass %d (a second pass in the same process shows the costs without first-launch effects)" % rep, flush=True)
    run()
