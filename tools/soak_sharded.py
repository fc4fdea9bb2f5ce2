"""Development aid: two processes on one GPU create / step / destroy a sharded handle over the peer-to-peer transport many
times (HIP IPC handles opened and closed every cycle); results must repeat and device memory must not creep."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def worker(rank, world, name, cycles, q):
    os.environ["FEMBRAIN_P2P"] = "1"
    from fembrain_amd import lib as fl
    from fembrain_amd.fem import FemIntegrator
    from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
    import torch
    L = fl.lib()
    n = 12
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    splits = np.array([n * r // world * n * n for r in range(world + 1)], np.int32)
    comm = C.c_void_p()
    fl.check(L.fb_comm_create_local(C.byref(comm), rank, world, name.encode(), 8 << 20, 0))
    ref, free0 = None, None
    for c in range(cycles):
        g = FemIntegrator(v, t, fixed, shard=(world, rank, splits, comm))
        assert g.transport() == fl.FB_XCH_P2P_FUSED
        g.set_exchange_mode([fl.FB_XCH_P2P, fl.FB_XCH_P2P_SUMS, fl.FB_XCH_P2P_FUSED][c % 3])
        g.set_uniform_force(1, -10000.0)
        it = g.do_timestep()
        qq = g.get_q_state()[0][3 * int(splits[rank]):3 * int(splits[rank + 1])].copy()
        g.close()
        if ref is None:
            ref = (it, qq)
        assert it == ref[0] and np.array_equal(qq, ref[1]), "cycle %d differs" % c
        free = torch.cuda.mem_get_info()[0]
        if c == 3:
            free0 = free
    q.put((rank, ref[0], (free - free0) / 2 ** 20))
    L.fb_comm_destroy(comm)


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, "/fembrain_soak_%d" % os.getpid(), 30, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(timeout=30)
    print(sorted(out))
