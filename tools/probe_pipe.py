"""development aid: the two forms of the PCG iteration on truth cubes (argv: nodes per side ...): two-launch (k_spmv + k_cg_fused) and the
pipelined persistent kernel: iteration counts, us per iteration, difference of the solutions, bitwise reproducibility across launch cuts"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("FEMBRAIN_PERSIST_MIN_WAVES", "1")
import numpy as np  # noqa: E402
from fembrain_amd import lib as fl  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402


def make(v, t, fixed, kind):
    os.environ.pop("FEMBRAIN_PCG_PERSIST", None)
    if kind == "two-launch":
        os.environ["FEMBRAIN_PCG_PERSIST"] = "0"
        return FemIntegrator(v, t, fixed, pcg_variant=fl.FB_PCG_MERGED)
    return FemIntegrator(v, t, fixed, pcg_variant=fl.FB_PCG_PERSISTENT)


for n in [int(a) for a in sys.argv[1:]] or [56]:
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    ref = None
    for kind in ("two-launch", "pipelined"):
        try:
            g = make(v, t, fixed, kind)
        except fl.FbError as e:
            print("n=%d %-18s n/a: %s" % (n, kind, e), flush=True)
            continue
        res = []
        for k in range(3):
            g.reset_to_rest()
            g.set_uniform_force(1, -10000.0)
            it = g.do_timestep()
            res.append((it, g.last.solve_seconds / max(it, 1) * 1e6))
        it2 = g.do_timestep()      # a second step from the deformed state
        q = g.get_q_state()[0]
        if ref is None:
            ref = q
        info = g.pcg_path()
        extra = ""
        if kind == "pipelined":
            # cut into launches of 7 iterations: bitwise the same solve
            g.reset_to_rest(); g.set_uniform_force(1, -10000.0); g.do_timestep(); qa = g.get_q_state()[0]
            os.environ["FEMBRAIN_PERSIST_MAX_RUN"] = "7"
            g.reset_to_rest(); g.set_uniform_force(1, -10000.0); itc = g.do_timestep(); qb = g.get_q_state()[0]
            os.environ.pop("FEMBRAIN_PERSIST_MAX_RUN")
            ta, tb = g.time_persist(3, 200), g.time_persist(3, 800)
            extra = " | cut/7: %d iterations, bitwise %s | launch-difference %.2f us/iteration" % (itc, np.array_equal(qa, qb), (tb - ta) / 600 * 1e6)
        print("n=%d %-18s iterations %s then %d  us/iter %s  maxrel vs two-launch %.2e  %s%s" % (
            n, kind, [r[0] for r in res], it2, ["%.2f" % r[1] for r in res], np.abs(q - ref).max() / np.abs(ref).max(), info, extra), flush=True)
        g.close()
