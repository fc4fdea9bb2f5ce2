"""Soak of fb_fem_resync_delta: a long random sequence of changes on a mid-size cube (the handle keeps the caller's node order, so every
plan array must equal a full re-sync's bit for bit), then the same on a handle with its own node order (pattern in caller ids equal,
a step within the solver's tolerance)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from fembrain_amd import lib as fl  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import apply_delta, cube_fixed_plane_i0, fixed_vertices_to_dofs, synthetic_cut, truth_cube  # noqa: E402

PLAN = ("bptr", "bcol", "blk_slot", "slice_off", "colidx", "slot_coff", "slot_ccnt", "contrib")


def device_plan(g, name):
    L = fl.lib()
    n = L.fb_fem_device_plan_get(g.h, name.encode(), None, 0)
    a = np.zeros(n, np.int32)
    assert L.fb_fem_device_plan_get(g.h, name.encode(), fl.iptr(a), n) == n
    return a


def random_delta(rng, cv, ct, scale):
    n_t, n_v = len(ct), len(cv)
    n_rem, n_chg, n_add, n_new = (int(x) for x in rng.integers(0, scale, size=4))
    ids = rng.permutation(n_t)[: n_rem + n_chg]
    rem = np.sort(ids[:n_rem]).astype(np.int32)
    chg = np.sort(ids[n_rem:]).astype(np.int32)
    base = rng.integers(0, n_v, size=n_new)
    new_xyz = cv[base] + rng.normal(size=(n_new, 3)) * 0.02
    allv = np.concatenate([cv, new_xyz])

    def tets_near(k):   # random elements among nodes that lie close together (a cut is local)
        out = np.empty((k, 4), np.int32)
        for i in range(k):
            while True:
                c = rng.integers(0, len(allv))
                cand = np.nonzero(np.abs(allv - allv[c]).max(axis=1) < 0.25)[0]
                if len(cand) < 4:
                    continue
                q = rng.choice(cand, 4, replace=False)
                p = allv[q]
                if abs(np.dot(np.cross(p[1] - p[0], p[2] - p[0]), p[3] - p[0])) > 1e-7:
                    out[i] = q
                    break
        return out
    return dict(removed=rem, changed_ids=chg, changed_nodes=tets_near(len(chg)), added=tets_near(n_add), new_xyz=new_xyz)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    rng = np.random.default_rng(7)
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    for renumber, label in ((fl.FB_RENUMBER_OFF, "caller's order"), (fl.FB_RENUMBER_ON, "own order")):
        g = FemIntegrator(v, t, fixed, renumber=renumber)
        ref = FemIntegrator(v, t, fixed, renumber=renumber)
        cv, ct = v, t
        t0 = time.time()
        paths = {}
        for k in range(rounds):
            d = random_delta(rng, cv, ct, 60) if k % 5 else synthetic_cut(cv, ct, axis=k % 3, where=float(rng.uniform(0.2, 0.8)), stride=7)[2]
            g.resync_delta(d, fixed, track=False)
            paths[g.resync_path()] = paths.get(g.resync_path(), 0) + 1
            cv, ct = apply_delta(cv, ct, d)
            ref.resync(cv, ct, fixed)
            if renumber == fl.FB_RENUMBER_OFF:
                for name in PLAN:
                    assert np.array_equal(device_plan(g, name), device_plan(ref, name)), (k, name)
            else:
                bp, bc = g.pattern()
                rp, rc = ref.pattern()
                assert np.array_equal(bp, rp) and np.array_equal(bc, rc), k
            if k % 6 == 5:
                its = []
                for h in (g, ref):
                    h.set_uniform_force(1, -50.0)
                    its.append(h.do_timestep())
                qa, qr = g.get_q_state()[0], ref.get_q_state()[0]
                if renumber == fl.FB_RENUMBER_OFF:
                    assert its[0] == its[1] and np.array_equal(qa, qr), (k, its)
                else:
                    assert np.abs(qa - qr).max() <= 1e-4 * max(np.abs(qr).max(), 1e-30), (k, its)
        print("%s: %d changes on a %d^3 cube -> %d nodes, %d tets; paths %s; %.1f s: ok" % (label, rounds, n, len(cv), len(ct), paths, time.time() - t0), flush=True)
        g.close()
        ref.close()


if __name__ == "__main__":
    main()
