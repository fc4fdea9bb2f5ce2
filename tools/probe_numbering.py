"""development aid: what the node NUMBERING of a mesh costs.  The same 1M-tet cube (argv[1] = nodes per side,
default 56) in (slab) its own grid order, (a) a random permutation of the node ids, (b) 5 % of the nodes moved to the end of the
list as CuttableMesh::cut appends the nodes it creates (VolMesh.cpp:1086-1091), (c) surface nodes first, then the interior, as
TetGen writes its outputs (data/models/blobtree/*.veg).  Per case: solver path, kernel, longest producer list, 16-bit column
words (from the SpMV bytes), us per PCG iteration, assembly us, re-sync ms, and the halo a contiguous-range partition gets at
2 / 4 / 8 ranks (neighbour ranks per rank, halo nodes per rank; host arithmetic on the element list)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from fembrain_amd import lib as fl  # noqa: E402
from fembrain_amd.fem import FemIntegrator  # noqa: E402
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube  # noqa: E402


def relabel(v, t, fixed_nodes, new_of_old):
    """node `old` becomes node new_of_old[old]"""
    v2 = np.empty_like(v)
    v2[new_of_old] = v
    return v2, new_of_old[t].astype(np.int32), np.sort(new_of_old[fixed_nodes]).astype(np.int32)


def halo_stats(t, n_nodes, n_ranks):
    splits = [n_nodes * i // n_ranks for i in range(n_ranks + 1)]
    owner = np.searchsorted(np.asarray(splits[1:]), np.arange(n_nodes), side="right")
    to = owner[t]                                   # owner of each corner
    nb, halo = [], []
    for r in range(n_ranks):
        mine = (to == r).any(axis=1)
        nodes = np.unique(t[mine])
        foreign = nodes[owner[nodes] != r]
        nb.append(len(np.unique(owner[foreign])))
        halo.append(len(foreign))
    return dict(ranks=n_ranks, max_neighbours=int(max(nb)), max_halo_nodes=int(max(halo)), mean_halo_nodes=float(np.mean(halo)))


def cases(n):
    v, t = truth_cube(n, n, n, 0.1)
    fx = cube_fixed_plane_i0(n, n)
    N = len(v)
    rng = np.random.default_rng(12345)
    yield "slab", v, t, fx
    yield "a_random_permutation", *relabel(v, t, fx, rng.permutation(N).astype(np.int64))
    moved = np.sort(rng.choice(N, N // 20, replace=False))
    keep = np.setdiff1d(np.arange(N), moved)
    new_of_old = np.empty(N, np.int64)
    new_of_old[keep] = np.arange(len(keep))
    new_of_old[rng.permutation(moved)] = len(keep) + np.arange(len(moved))
    yield "b_cut_appended_5pct", *relabel(v, t, fx, new_of_old)
    ijk = np.stack(np.unravel_index(np.arange(N), (n, n, n)), axis=1)
    surf = ((ijk == 0) | (ijk == n - 1)).any(axis=1)
    order = np.concatenate([np.nonzero(surf)[0], np.nonzero(~surf)[0]])
    new_of_old = np.empty(N, np.int64)
    new_of_old[order] = np.arange(N)
    yield "c_surface_first", *relabel(v, t, fx, new_of_old)
    if os.environ.get("PROBE_DELAUNAY", "1") != "0":
        # (d) an UNSTRUCTURED mesh: Delaunay tetrahedra of a jittered grid (no grid order to find again), nodes in random order
        from fembrain_amd.meshgen import delaunay_jittered
        pts, tt, fxd = delaunay_jittered(max(8, int(round(n * 0.8))), max_edge=float(os.environ["PROBE_MAX_EDGE"]) if os.environ.get("PROBE_MAX_EDGE") else None)
        yield "d_delaunay_jittered_random_order", pts, tt, fxd


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
    out = []
    only = os.environ.get("PROBE_ONLY")   # e.g. "d": the cases whose name starts with it
    for name, v, t, fx in cases(n):
        if only and not name.startswith(only):
            continue
        fixed = fixed_vertices_to_dofs(fx)
        t0 = time.time()
        g = FemIntegrator(v, t, fixed)
        create_s = time.time() - t0
        row = dict(case=name, nodes=len(v), tets=len(t), create_s=round(create_s, 3))
        its, us = [], []
        for _ in range(3):
            g.reset_to_rest()
            g.set_uniform_force(1, -10.0 if name.startswith("d_") else -10000.0)
            try:
                it = g.do_timestep()
            except fl.FbError as e:       # (a sliver mesh may not converge in 10,000 iterations: the time per iteration is what is asked)
                row["note"] = str(e)[:80]
                it = g.last.cg_iterations
            its.append(it)
            us.append(g.last.solve_seconds / max(it, 1) * 1e6)
        row["renumbering"] = g.renumbering()
        row.update(iterations=its, us_per_iteration=[round(u, 2) for u in us], path=g.pcg_path(), persist=g.persist_info(), gather=g.persist_gather(),
                   spmv_mb=round(g.spmv_bytes() / 1e6, 1), spmv_us=round(g.time_spmv(50) * 1e6, 2),
                   assembly_us=round(g.time_assembly(10) * 1e6, 1), assembly_kernel=int(g._L.fb_fem_assembly_kernel(g.h)))
        # widths of the SELL slices (the element-major assembly takes up to 31 slots; wider slices went to the slot-major kernel)
        cnt = g._L.fb_fem_device_plan_get(g.h, b"slice_off", None, 0)
        so = np.zeros(cnt, np.int32)
        g._L.fb_fem_device_plan_get(g.h, b"slice_off", fl.iptr(so), cnt)
        w = np.diff(so)
        if os.environ.get("PROBE_DUMP"):
            np.save(os.path.join(os.environ["PROBE_DUMP"], "widths_%s.npy" % row["case"][:1]), w)
        row["slice_widths"] = dict(slices=int(len(w)), max=int(w.max()), mean=float(w.mean()), wider_than_31=int((w > 31).sum()), p50=int(np.percentile(w, 50)), p99=int(np.percentile(w, 99)))
        t0 = time.time()
        for _ in range(5):
            g.resync(v, t, fixed)
        row["resync_ms_incl_upload"] = round((time.time() - t0) / 5 * 1e3, 2)
        row["halo"] = [halo_stats(t, len(v), r) for r in (2, 4, 8)]
        g.close()
        print(json.dumps(row), flush=True)
        out.append(row)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
