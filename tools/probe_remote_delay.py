"""the N > 1 solver forms under a DELAYED "xGMI" (FEMBRAIN_REMOTE_DELAY_US: every remote signal is raised and acted on
that much later, fembrain_amd/csrc/p2p_device.hip.h remote_delay), rehearsed by `world` processes on CU shares of the one GPU.  Per cube,
rank count and delay: us per PCG iteration of the sharded persistent solver (k_pcg_pipe / k_pcg_pipe2 with SHARD) under its default
relief and with none, and of the two-launch iteration with the fused peer-to-peer exchange (FB_XCH_P2P_FUSED).
    python tools/probe_remote_delay.py out.json [n ...]      (default cubes: 56, 70)"""
import ctypes as C
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, name, n, steps, q, form, delay):
    try:
        os.environ["FEMBRAIN_CU_MASK"] = "%d:%d" % (rank * (256 // world // 32 * 32), 256 // world // 32 * 32)
        os.environ["FEMBRAIN_PERSIST_TIMEOUT_MS"] = "2000"
        os.environ["FEMBRAIN_P2P"] = "1"
        os.environ["FEMBRAIN_REMOTE_DELAY_US"] = str(delay)
        if form.startswith("persist"):
            os.environ["FEMBRAIN_SHARDED_PERSIST"] = "1"
            if form == "persist_relief0":
                os.environ["FEMBRAIN_SHARD_RELIEF"] = "0"
        else:
            os.environ["FEMBRAIN_XCH_MODE"] = "4"
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
        q.put((rank, its, solve, g.pcg_path()["kernel"], g.pcg_path()["fallbacks"], int(g.last.pcg_path)))
        g.close()
        L.fb_comm_destroy(comm)
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e), 0.0, "", 0, -1))


if __name__ == "__main__":
    out_path = sys.argv[1]
    cubes = [int(a) for a in sys.argv[2:]] or [56, 70]
    ctx = mp.get_context("spawn")
    rows = []
    for n in cubes:
        for world in (2, 4):
            for delay in (0, 1, 2, 5):
                row = {"cube": n, "tets": 6 * (n - 1) ** 3, "ranks": world, "cus_per_rank": 256 // world // 32 * 32, "remote_delay_us": delay}
                for form in ("persist", "persist_relief0", "p2p_fused"):
                    q = ctx.Queue()
                    ps = [ctx.Process(target=worker, args=(r, world, "/fb_probe_rd_%d_%d" % (os.getpid(), len(rows)), n, 2, q, form, delay)) for r in range(world)]
                    for p in ps:
                        p.start()
                    res = sorted(q.get(timeout=600) for _ in range(world))
                    for p in ps:
                        p.join()
                    if any(isinstance(r[1], str) for r in res):
                        row[form] = {"error": [r[1] for r in res if isinstance(r[1], str)][0][:160]}
                        continue
                    its = res[0][1]
                    row[form] = {"us_per_iteration": round(max(r[2] for r in res) / sum(its) * 1e6, 2), "iterations": its, "kernel": res[0][3] or "k_spmv + k_cg_fused",
                                 "fallbacks": max(r[4] for r in res), "persistent": all(r[5] == 1 for r in res)}
                print(json.dumps(row), flush=True)
                rows.append(row)
                json.dump(rows, open(out_path, "w"), indent=1)
