"""Two ranks on one GPU: time of a collective fb_fem_resync_sharded and the SpMV byte count of a sharded handle
(2-byte column deltas where they fit).  usage: python tools/probe_sharded_resync.py [n=71] [world=2] [own|whole]
(own: every rank passes its own elements; whole: every rank passes the whole mesh and the device keeps its share)"""
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def worker(rank, world, name, n, q, whole=False):
    from fembrain_amd import lib as fl
    from fembrain_amd.fem import FemIntegrator
    from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
    L = fl.lib()
    comm = C.c_void_p()
    fl.check(L.fb_comm_create_local(C.byref(comm), rank, world, name.encode(), 64 << 20, 0))
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    splits = np.array([(n * r // world) * n * n for r in range(world + 1)], np.int32)
    own = ((t >= splits[rank]) & (t < splits[rank + 1])).any(axis=1)
    to = np.ascontiguousarray(t if whole else t[own])
    t0 = time.perf_counter()
    g = FemIntegrator(v, to, fixed, shard=(world, rank, splits, comm))
    create_ms = (time.perf_counter() - t0) * 1e3
    ms = []
    for _ in range(4):
        t0 = time.perf_counter()
        g.resync(v, to, fixed, node_splits=splits)
        ms.append((time.perf_counter() - t0) * 1e3)
    g.set_uniform_force(1, -3000.0)
    its = g.do_timestep()
    q.put((rank, int(own.sum()), create_ms, ms, g.num_blocks(), g.spmv_bytes(), its, g.last.solve_seconds))
    g.close()
    L.fb_comm_destroy(comm)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 71
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    whole = len(sys.argv) > 3 and sys.argv[3] == "whole"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/fembrain_probe_resync_%d" % os.getpid()
    ps = [ctx.Process(target=worker, args=(r, world, name, n, q, whole)) for r in range(world)]
    for p in ps:
        p.start()
    for _ in ps:
        rank, ne, cms, ms, nb, by, its, solve = q.get(timeout=600)
        print("rank %d: %d own tets, create %.1f ms, resync %s ms, %d blocks, spmv %.1f MB (%.1f B/block), %d iterations, solve %.3f s"
              % (rank, ne, cms, ["%.1f" % m for m in ms], nb, by / 1e6, by / nb, its, solve), flush=True)
    for p in ps:
        p.join(timeout=60)
