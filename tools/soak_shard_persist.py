"""Soak of the sharded persistent solver on ONE GPU: `world` processes on disjoint CU shares step the truth cube n^3 `steps` times under
a load that changes every step, twice over; the two passes must agree BIT FOR BIT on every rank's state after every step (a race in
the hand-overs -- a halo row read before it was copied, a sum taken from the wrong parity -- shows up as a difference), no launch may
time out, and the iteration counts of the first ten steps must be those of the two-launch sharded iteration to max(2, 1 %) (later the
two trajectories drift apart like any two solves of this model that stop at 1e-6 -- 10 % either way after 40 loaded steps -- which says
nothing about either solver).
   python tools/soak_shard_persist.py [n=40] [world=2] [steps=60]"""
import ctypes as C
import hashlib
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, name, n, steps, q, persist):
    os.environ["FEMBRAIN_CU_MASK"] = "%d:%d" % (rank * (256 // world // 32 * 32), 256 // world // 32 * 32)
    os.environ["FEMBRAIN_P2P"] = "1"
    if persist:
        os.environ["FEMBRAIN_SHARDED_PERSIST"] = "1"
    from fembrain_amd import lib as fl
    from fembrain_amd.fem import FemIntegrator
    from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
    L = fl.lib()
    comm = C.c_void_p()
    fl.check(L.fb_comm_create_local(C.byref(comm), rank, world, name.encode(), 64 << 20, 0))
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    splits = np.array([(n * r // world) * n * n for r in range(world + 1)], np.int32)
    g = FemIntegrator(v, t, fixed, shard=(world, rank, splits, comm))
    its, digests = [], []
    for k in range(steps):
        f = np.zeros(g.r)
        f[1::3] = -2000.0 * (1.0 + 0.5 * np.sin(0.37 * k))
        f[0::3] = 150.0 * np.sin(np.arange(len(v)) + k)
        g.set_external_forces(f)
        its.append(g.do_timestep())
        qq, vv, _ = g.get_q_state()
        lo, hi = 3 * int(splits[rank]), 3 * int(splits[rank + 1])
        digests.append(hashlib.sha256(qq[lo:hi].tobytes() + vv[lo:hi].tobytes()).hexdigest()[:16])
    q.put((rank, its, digests, g.pcg_path()))
    g.close()
    L.fb_comm_destroy(comm)


def run(n, world, steps, persist, tag):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, "/fb_soak_sp_%d_%s" % (os.getpid(), tag), n, steps, q, persist)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(world))
    for p in ps:
        p.join()
    return res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    a = run(n, world, steps, 1, "a")
    b = run(n, world, steps, 1, "b")
    ref = run(n, world, steps, 0, "r")
    ok = True
    for ra, rb in zip(a, b):
        same = ra[2] == rb[2] and ra[1] == rb[1]
        first = next((k for k in range(steps) if ra[2][k] != rb[2][k]), None)
        print("rank %d: %s, %d launches, %d fallbacks; passes %s%s" % (ra[0], ra[3]["kernel"], ra[3]["launches"], ra[3]["fallbacks"], "identical" if same else "DIFFER",
                                                                     "" if same else " from step %s" % first), flush=True)
        ok = ok and same and ra[3]["fallbacks"] == 0 and "shard" in ra[3]["kernel"]
    worst = max(abs(x - y) / max(y, 1) for x, y in zip(a[0][1][:10], ref[0][1][:10]))
    print("iterations: persistent %d..%d, two-launch %d..%d, largest relative difference over the first ten steps %.4f" % (min(a[0][1]), max(a[0][1]), min(ref[0][1]), max(ref[0][1]), worst))
    bad = [(k, x, y) for k, (x, y) in enumerate(zip(a[0][1][:10], ref[0][1][:10])) if abs(x - y) > max(2, 0.01 * y)]
    if bad:
        print("steps whose iteration counts differ by more than max(2, 1 %) (step, persistent, two-launch):", bad[:12])
    ok = ok and not bad
    print("soak", "ok" if ok else "FAILED")
    sys.exit(0 if ok else 1)
