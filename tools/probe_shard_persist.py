"""Rehearsal timing of the sharded persistent solver on ONE GPU: `world` processes, each confined to its own share of the CUs
(FEMBRAIN_CU_MASK), step the truth cube n^3 cut into slabs; prints the device time per PCG iteration of the persistent launches (HIP
events around every launch) and, beside it, the solve time per iteration of the two-launch sharded iteration in its exchange modes on
the same CU shares.  "Remote" stores land in the same HBM here, so this measures the protocol (drain -> counter -> proxy copy -> flag),
not xGMI.   python tools/probe_shard_persist.py [n=56] [world=2] [steps=4]"""
import ctypes as C
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, name, n, steps, q, persist, mode):
    os.environ["FEMBRAIN_CU_MASK"] = "%d:%d" % (rank * (256 // world // 32 * 32), 256 // world // 32 * 32)
    os.environ["FEMBRAIN_PERSIST_TIMEOUT_MS"] = "2000"
    os.environ["FEMBRAIN_P2P"] = "1"
    if persist:
        os.environ["FEMBRAIN_SHARDED_PERSIST"] = "1"
    else:
        os.environ["FEMBRAIN_XCH_MODE"] = str(mode)
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
    its, solve = [], 0.0
    for k in range(steps + 1):
        g.set_uniform_force(1, -10000.0)
        it = g.do_timestep()
        if k:
            its.append(it)
            solve += g.last.solve_seconds
    st = g.persist_stats() if persist else None
    q.put((rank, its, solve, st, g.pcg_path()["kernel"], g.pcg_path()["fallbacks"]))
    g.close()
    L.fb_comm_destroy(comm)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    ctx = mp.get_context("spawn")
    for persist, mode in ((1, 0), (0, 2), (0, 4)):
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(r, world, "/fb_probe_sp_%d_%d%d" % (os.getpid(), persist, mode), n, steps, q, persist, mode)) for r in range(world)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=300) for _ in range(world))
        for p in ps:
            p.join()
        r0 = res[0]
        line = "cube %d^3, %d ranks on %d CUs each: %s  iterations %s  solve %.1f us/iteration" % (
            n, world, 256 // world // 32 * 32, "sharded persistent" if persist else "two-launch, exchange mode %d" % mode, r0[1], max(r[2] for r in res) / sum(r0[1]) * 1e6)
        if persist:
            line += "  | launches: " + ", ".join("rank %d %.2f us/iteration (%d launches, kernel %s, %d fallbacks)" % (r[0], r[3][1] / max(r[3][2], 1) * 1e6, r[3][0], r[4], r[5]) for r in res)
        print(line, flush=True)
