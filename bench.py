#!/usr/bin/env python3
"""Headline benchmark: FEM steps/s (element rebuild + assembly + Jacobi-PCG) on the synthetic 1M-tet cantilever,
and BlobTree field Mvoxels/s on a 256^3 grid (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One JSON line on rank 0.  A "step" is one pass of the hot path: fb_fem_rebuild_elements (per-element rest state,
the cutting path's K0 rebuild), corotational assembly of Keff/rhs and the PCG solve to the reference tolerance
(eps 1e-6, max 10000), followed by the state update -- all inputs resident in HBM.  For N > 1 the SAME mesh is
slab-decomposed over the ranks (strong scaling): one process per GPU, RCCL all-reduce for the PCG dots and
ncclSend/Recv halo exchange of the search direction.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (nodes per side, text)
    "cube27": (27, "truth cube 27^3 nodes / 105,456 tets (BASELINE config 2 canonical mesh)"),
    "cube56": (56, "truth cube 56^3 nodes / 998,250 tets, plane i=0 clamped, -10000 per y DOF, per-step re-assembly (BASELINE config 4)"),
    "cube111": (111, "truth cube 111^3 nodes / 7,986,000 tets (BASELINE config 5 mesh)"),
    # between the two: the sizes the two-row persistent solver (k_pcg_pipe2) takes, 13..24 slices per CU
    "cube64": (64, "truth cube 64^3 nodes / 1,500,282 tets (16 slices per CU: two-row persistent solver)"),
    "cube73": (73, "truth cube 73^3 nodes / 2,239,488 tets (24 slices per CU: the largest system the persistent solver takes)"),
    # BASELINE config 2 on a reference model: ventricle.blob polygonized on the device at cellsize 0.115
    "ventricle": (0, "ventricle.blob (17 primitives) -> tetrahedral polygonizer at cellsize 0.115 -> 107,820 tets, lowest 5 % of the nodes "
                     "in y clamped, -10000 per y DOF (BASELINE config 2)"),
}


def workload_mesh(name, device=0):
    """(vertices, tets, constrained DOFs) of a workload; the BlobTree model is polygonized by the HIP path itself."""
    from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
    n = WORKLOADS[name][0]
    if n:
        v, t = truth_cube(n, n, n, 0.1)
        return v, t, fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    from fembrain_amd.blobtree import read_blob
    from fembrain_amd.poly import GpuPoly
    p = GpuPoly(read_blob(os.path.join(ROOT, "tests", "golden", "blob", "ventricle.blob")), device=device)
    xyz, tets = p.run_tetrahedralizer(0.115)
    p.close()
    v, t = xyz.astype(np.float64), tets.astype(np.int32)
    ycut = np.sort(v[:, 1])[len(v) // 20]
    return v, t, fixed_vertices_to_dofs(np.nonzero(v[:, 1] <= ycut)[0])


def cpu_baseline(mesh, cg_iterations, sample_iters=40):
    """The reference's CPU path on the same workload, 1 core (it is serial: Deformable.cpp:182).  When the reference build
    oracle/_ref/libfem_ref.so travelled with the snapshot ("reference": the reference's own CorotationalLinearFEM,
    SparseMatrix and CGSolver objects driven through its step sequence) it is timed, otherwise the plain-C restatement
    oracle/fem_oracle.c ("port").  Bounded sample: one step cut off after 1 PCG iteration and one cut off after
    1 + sample_iters; the difference prices an iteration, the rest is assembly + system algebra; the step time is
    extrapolated with the iteration count the GPU run needed.  Set-up is not timed."""
    from oracle import pyoracle
    v, t, fixed = mesh

    def sample(cls):
        o = cls(v, t)
        o.integrator(fixed)
        f = np.zeros(o.r)
        f[1::3] = -10000.0
        o.set_external_forces(f)
        t0 = time.perf_counter()
        o.step(cg_maxiter=1)
        t1 = time.perf_counter()
        o.step(cg_maxiter=1 + sample_iters)
        t2 = time.perf_counter()
        o.close()
        t_it = max(((t2 - t1) - (t1 - t0)) / sample_iters, 1e-9)
        t_asm = max((t1 - t0) - t_it, 0.0)
        return t_asm, t_it

    kind, cls = "port", pyoracle.OrcFem
    if os.environ.get("FEMBRAIN_BENCH_CPU_KIND") != "port":
        try:
            pyoracle._load("ref")
            kind, cls = "reference", pyoracle.RefFem
        except Exception:
            pass
    t_asm, t_it = sample(cls)
    step_s = t_asm + cg_iterations * t_it
    what = ("oracle/_ref/libfem_ref.so = the reference's VegaFEM translation units (CorotationalLinearFEM, SparseMatrix, CGSolver) built "
            "from /root/reference by oracle/Makefile, driven through VolumeConservingIntegrator's step sequence") if kind == "reference" \
        else "oracle/fem_oracle.c (plain-C restatement)"
    return {"value": 1.0 / step_s, "unit": "steps/s", "cores": 1, "kind": kind,
            "sample": "%s on the same mesh: assembly + system algebra %.2f s, %d PCG iterations at %.2f ms each, extrapolated to the %d "
                      "iterations per step of the GPU run; set-up excluded" % (what, t_asm, sample_iters, t_it * 1e3, cg_iterations)}


def cpu_full_step(mesh):
    """ONE full step of the reference's CPU path (oracle/_ref/libfem_ref.so, else the C restatement) on `mesh`, 1 core, from rest under
    the reference load, PCG to its tolerance -- no extrapolation (2.3-2.6 s at 105k tets)."""
    from oracle import pyoracle
    v, t, fixed = mesh
    kind, cls = "port", pyoracle.OrcFem
    if os.environ.get("FEMBRAIN_BENCH_CPU_KIND") != "port":
        try:
            pyoracle._load("ref")
            kind, cls = "reference", pyoracle.RefFem
        except Exception:
            pass
    o = cls(v, t)
    o.integrator(fixed)
    f = np.zeros(o.r)
    f[1::3] = -10000.0
    o.set_external_forces(f)
    t0 = time.perf_counter()
    it = abs(o.step())
    dt = time.perf_counter() - t0
    o.close()
    rec = {"value": 1.0 / dt, "unit": "steps/s", "cores": 1, "kind": kind, "cg_iterations": int(it),
           "sample": "one whole step from rest (assembly + system algebra + %d PCG iterations), %.2f s; set-up excluded" % (it, dt)}
    if kind == "port":
        rec["why_port"] = "oracle/_ref/libfem_ref.so (the reference's own translation units) did not travel with this snapshot; oracle/fem_oracle.c is its restatement"
    return rec


def small_leg(name, device, prec, cpu, steps=5):
    """BASELINE configs 1-2 as driver-timed legs (never part of `value`): the 27^3 truth cube (105,456 tets) and a BlobTree model
    polygonized on the device to ~100k tets and handed to the FEM handle without a host hop (fb_fem_create_from_poly).  Steps from
    the rest state under the reference load (every step the same system: `value`), and the steps of the loaded trajectory."""
    from fembrain_amd import lib as fl
    from fembrain_amd.fem import FemIntegrator
    from fembrain_amd.meshgen import fixed_vertices_to_dofs
    import torch
    poly = None
    if WORKLOADS[name][0]:
        v, t, fixed = workload_mesh(name, device)
        g = FemIntegrator(v, t, fixed, matrix_precision=prec, device=device)
    else:
        from fembrain_amd.blobtree import read_blob
        from fembrain_amd.poly import GpuPoly
        poly = GpuPoly(read_blob(os.path.join(ROOT, "tests", "golden", "blob", "ventricle.blob")), device=device)
        xyz, tets = poly.run_tetrahedralizer(0.115)
        v, t = xyz.astype(np.float64), tets.astype(np.int32)
        ycut = np.sort(v[:, 1])[len(v) // 20]
        fixed = fixed_vertices_to_dofs(np.nonzero(v[:, 1] <= ycut)[0])
        g = FemIntegrator.from_poly(poly, fixed, matrix_precision=prec, device=device)

    def step():
        g.rebuild_elements()
        g.set_uniform_force(1, -10000.0)
        return g.do_timestep()
    step()
    tot, its, solve, asm = 0.0, [], 0.0, 0.0
    for _ in range(steps):
        g.reset_to_rest()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        its.append(step())
        torch.cuda.synchronize()
        tot += time.perf_counter() - ts
        solve += g.last.solve_seconds
        asm += g.last.assembly_seconds
    g.reset_to_rest()
    torch.cuda.synchronize()
    ts = time.perf_counter()
    traj = [step() for _ in range(steps)]
    torch.cuda.synchronize()
    traj_dt = time.perf_counter() - ts
    path = g.pcg_path()
    out = {"workload": WORKLOADS[name][1], "nodes": int(len(v)), "tets": int(len(t)), "steps": steps, "value": steps / tot, "unit": "steps/s",
           "value_is": "steps from the rest state under the reference load (the same system every step)", "ms_per_step": tot / steps * 1e3,
           "cg_iterations": [int(i) for i in its], "us_per_cg_iteration": solve / max(sum(its), 1) * 1e6, "assembly_ms_per_step": asm / steps * 1e3,
           "trajectory_steps_per_s": steps / traj_dt, "trajectory_cg_iterations": [int(i) for i in traj],
           "pcg_kernel": path["kernel"] or ("k_spmv_split + k_cg_fused (hipGraph of 30 iterations)" if len(v) < 512 * 64 else "k_spmv + k_cg_fused"),
           "pcg_path_last_step": int(g.last.pcg_path), "max_producers_per_workgroup": path["max_producers"],
           "created_from": "fb_fem_create_from_poly (tet mesh left on the device by the polygonizer)" if poly is not None else "fb_fem_create",
           "renumbered": bool(g.renumbering()[0])}
    g.close()
    if poly is not None:
        poly.close()
    if cpu:
        try:
            out["cpu_baseline"] = cpu_full_step((v, t, fixed))
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline_error"] = repr(e)
    return out


def numbering_leg(device, prec):
    """the headline mesh in a RANDOM node order (what a caller's numbering may be after cuts: CuttableMesh::cut
    appends its new nodes at the end of the list).  The handle renumbers internally (fembrain_amd/csrc/renumber.h); reported: the
    kernel it ends up with, us per PCG iteration, the re-sync time including the renumbering."""
    from fembrain_amd.fem import FemIntegrator
    from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
    import torch
    n = WORKLOADS["cube56"][0]
    v0, t0 = truth_cube(n, n, n, 0.1)
    m = np.random.default_rng(12345).permutation(len(v0))
    v = np.empty_like(v0)
    v[m] = v0
    t = np.ascontiguousarray(m[t0].astype(np.int32))
    fixed = fixed_vertices_to_dofs(np.sort(m[cube_fixed_plane_i0(n, n)]))
    g = FemIntegrator(v, t, fixed, matrix_precision=prec, device=device)
    its, solve = [], 0.0
    for k in range(3):
        g.reset_to_rest()
        g.set_uniform_force(1, -10000.0)
        it = g.do_timestep()
        if k:
            its.append(it)
            solve += g.last.solve_seconds
    rs = []
    for _ in range(3):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        g.resync(v, t, fixed)
        torch.cuda.synchronize()
        rs.append((time.perf_counter() - ts) * 1e3)
    on, sc, si = g.renumbering()
    path = g.pcg_path()
    g.close()
    return {"workload": "the 56^3 truth cube with its node ids randomly permuted", "renumbered": bool(on), "widest_element_caller_order": sc,
            "widest_element_internal_order": si, "pcg_kernel": path["kernel"], "max_producers_per_workgroup": path["max_producers"],
            "cg_iterations": [int(i) for i in its], "us_per_cg_iteration": solve / max(sum(its), 1) * 1e6, "resync_ms": min(rs),
            "resync_ms_all": [round(x, 3) for x in rs]}


def unstructured_leg(device, prec, m=45):
    """an UNSTRUCTURED mesh -- Delaunay tetrahedra of a jittered 45^3 lattice in random node order (606k tets, hull nodes
    with 40-60 neighbours; fembrain_amd.meshgen.delaunay_jittered) -- the class of every TetGen mesh the reference ships and of every mesh
    after a cut.  Reported: the solver kernel and its us per PCG iteration, how the gathered vector is laid out and why (cache lines per
    slot), one assembly, the slice widths."""
    from fembrain_amd import lib as fl
    from fembrain_amd.fem import FemIntegrator
    from fembrain_amd.meshgen import delaunay_jittered, fixed_vertices_to_dofs
    v, t, fv = delaunay_jittered(m)
    g = FemIntegrator(v, t, fixed_vertices_to_dofs(fv), matrix_precision=prec, device=device)
    its, solve = [], 0.0
    for k in range(3):
        g.reset_to_rest()
        g.set_uniform_force(1, -100.0)
        it = g.do_timestep()
        if k:
            its.append(it)
            solve += g.last.solve_seconds
    asm = g.time_assembly(10)
    L = g._L
    cnt = L.fb_fem_device_plan_get(g.h, b"slice_off", None, 0)
    so = np.zeros(cnt, np.int32)
    L.fb_fem_device_plan_get(g.h, b"slice_off", fl.iptr(so), cnt)
    w = np.diff(so)
    xyz, lp, lr = g.persist_gather()
    on, sc, si = g.renumbering()
    path = g.pcg_path()
    out = {"workload": "Delaunay tetrahedra of a jittered %d^3 lattice, nodes in random order: %d nodes, %d tets" % (m, len(v), len(t)),
           "renumbered": bool(on), "widest_element_internal_order": si, "pcg_kernel": path["kernel"], "max_producers_per_workgroup": path["max_producers"],
           "helper_tasks": int(L.fb_fem_persist_helpers(g.h)),
           "gathered_vector": {"layout": "node by node (24-byte records)" if xyz else "three planes", "cache_lines_per_slot_planes": round(lp, 1),
                               "cache_lines_per_slot_records": round(lr, 1)},
           "cg_iterations": [int(i) for i in its], "us_per_cg_iteration": solve / max(sum(its), 1) * 1e6,
           "assembly_us": asm * 1e6, "assembly_kernel": int(L.fb_fem_assembly_kernel(g.h)), "assembly_wide_slices": int(L.fb_fem_assembly_wide_slices(g.h)),
           "slice_widths": {"slices": int(len(w)), "max": int(w.max()), "mean": round(float(w.mean()), 2), "wider_than_31": int((w > 31).sum())},
           "spmv_mb": g.spmv_bytes() / 1e6}
    g.close()
    return out


def block_jacobi_leg(device, prec, steps=5):
    """the headline mesh with the OPT-IN 3x3 block-Jacobi preconditioner (FB_PCG_BLOCK_JACOBI, not the reference's
    preconditioner, outside every parity claim and never part of `value`) inside the same persistent kernel (k_pcg_pipe<.., BJ>):
    steps from the rest state under the reference load, as `value` is."""
    from fembrain_amd import lib as fl
    from fembrain_amd.fem import FemIntegrator
    import torch
    v, t, fixed = workload_mesh("cube56", device)
    g = FemIntegrator(v, t, fixed, matrix_precision=prec, device=device, pcg_variant=fl.FB_PCG_BLOCK_JACOBI)

    def step():
        g.rebuild_elements()
        g.set_uniform_force(1, -10000.0)
        return g.do_timestep()
    step()
    tot, its, solve = 0.0, [], 0.0
    for _ in range(steps):
        g.reset_to_rest()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        its.append(step())
        torch.cuda.synchronize()
        tot += time.perf_counter() - ts
        solve += g.last.solve_seconds
    path, last_path = g.pcg_path(), int(g.last.pcg_path)
    g.close()
    return {"workload": WORKLOADS["cube56"][1], "preconditioner": "inverse 3x3 diagonal blocks (opt-in; the reference and `value` use 1/diag)",
            "parity": "outside the parity claim: same system, same tolerance, another preconditioner (tests compare the solutions at 1e-4)",
            "steps": steps, "value": steps / tot, "unit": "steps/s", "ms_per_step": tot / steps * 1e3, "cg_iterations": [int(i) for i in its],
            "us_per_cg_iteration": solve / max(sum(its), 1) * 1e6, "pcg_kernel": path["kernel"], "pcg_path_last_step": last_path}


def cut_resync_leg(device, prec):
    """VERDICT r3 item 7 / SURVEY 8f-3: the re-sync after a cut on the headline mesh, from a DESCRIPTION of the change (fb_fem_resync_delta:
    the mesh stays on the device, the plan's sorted pair list is updated) against fb_fem_resync with the whole new mesh.  First a real
    cut (meshgen.synthetic_cut: every element crossing a plane split in four on a new node; the mesh grows, so buffers grow with it), then
    changes of constant size -- 1 % of the elements removed and appended again, mirrored -- that time the re-sync itself."""
    from fembrain_amd import lib as fl
    from fembrain_amd.fem import FemIntegrator
    from fembrain_amd.meshgen import apply_delta, synthetic_cut
    import torch
    v, t, fixed = workload_mesh("cube56", device)
    names = {fl.FB_RESYNC_FULL: "full", fl.FB_RESYNC_DELTA_MERGED: "plan updated from the plan", fl.FB_RESYNC_DELTA_REBUILT: "full builder from the device copy of the mesh"}
    g = FemIntegrator(v, t, fixed, matrix_precision=prec, device=device, expect_cuts=True)   # (the caller of fb_fem_resync_delta says it will cut)
    ref = FemIntegrator(v, t, fixed, matrix_precision=prec, device=device)

    def both(d, v2, t2):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        g.resync_delta(d, fixed, track=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - ts) * 1e3
        ts = time.perf_counter()
        ref.resync(v2, t2, fixed)
        torch.cuda.synchronize()
        return dt, (time.perf_counter() - ts) * 1e3, names[g.resync_path()]
    cv, ct = v, t
    cuts = []
    for k in range(2):
        v2, t2, d = synthetic_cut(cv, ct, axis=1 + k, where=0.23 + 0.2 * k)
        dt, dr, path = both(d, v2, t2)
        cuts.append({"removed": int(len(d["removed"])), "changed": int(len(d["changed_ids"])), "added": int(len(d["added"])), "new_nodes": int(len(d["new_xyz"])),
                     "tets_after": int(len(t2)), "delta_ms": dt, "delta_path": path, "full_resync_ms": dr})
        cv, ct = v2, t2
    small = []
    for k in range(4):
        x = cv[:, 1][ct]
        c = cv[:, 1].min() + (0.31 + 0.07 * k) * (cv[:, 1].max() - cv[:, 1].min())
        hit = np.nonzero((x.min(axis=1) < c) & (x.max(axis=1) > c))[0].astype(np.int32)
        hit = hit[:: max(1, len(hit) // (len(ct) // 100))][: len(ct) // 100]
        d = dict(removed=hit, changed_ids=np.zeros(0, np.int32), changed_nodes=np.zeros(0, np.int32), added=ct[hit][:, [1, 0, 2, 3]], new_xyz=np.zeros(0))
        v2, t2 = apply_delta(cv, ct, d)
        dt, dr, path = both(d, v2, t2)
        small.append((dt, dr, path, int(len(hit))))
        cv, ct = v2, t2
    its = []
    for h in (g, ref):
        h.set_uniform_force(1, -10000.0)
        its.append(int(h.do_timestep()))
    out = {"workload": WORKLOADS["cube56"][1] + ", cut twice", "cuts": cuts,
           "one_percent_change": {"elements_removed_and_appended": small[-1][3], "tets": int(len(ct)), "delta_ms": min(x[0] for x in small[1:]),
                                  "delta_ms_all": [round(x[0], 3) for x in small], "delta_path": small[-1][2],
                                  "full_resync_ms": min(x[1] for x in small[1:]), "full_resync_ms_all": [round(x[1], 3) for x in small]},
           "first_step_cg_iterations_delta_vs_full": its,
           "note": "host wall clock around the C call incl. the Python wrapper; the first cut of a handle in the caller's node order appends nodes, which "
                   "makes the full builder choose a new node order (from the device copy of the mesh); later changes update the pair list"}
    g.close()
    ref.close()
    return out


def field_bench(device, cpu=True):
    """256^3 sweep + classify + tetrahedralize of sphere.blob (BASELINE config 3); returns extra JSON keys."""
    from fembrain_amd.poly import GpuPoly, sphere_blob
    blob = sphere_blob()
    lower, cell, dims = (-0.5, -0.5, -0.5), 1.0 / 254.0, (256, 256, 256)
    p = GpuPoly(blob, device=device)
    p.sweep_grid(lower, cell, dims)
    c = p.classify()
    p.tetrahedralize()
    sweep_s, pipe_s = p.time_pipeline(10)
    st = p.time_stages(10)
    grid_s = p.time_grid(10)   # sweep + the float4 (x, y, z, f) grid the API returns on request (16 B per point)
    p.surface()
    surf_s = p.time_surface(10)
    npts = dims[0] * dims[1] * dims[2]
    # roofline of the field half: the dominant kernel is k_tet_elements, a pure store stream -- 6 tets x 16 B per
    # included cell -- plus the two bit masks and scan bases it reads (2 x 1 bit + 2 x 4 B per 64 points); the whole pipeline moves
    # 16 B per point (sweep) + 12 B per tet-mesh vertex + 96 B per included cell + ~3 B per point of masks and scans
    n_inc, n_tv = int(c.n_included_cells), int(p.counts.n_tet_vertices)
    # PMC traffic of the dominant kernel from the committed profile, only while the kernel sources are the profiled ones
    ftraffic, fnote = None, "no PMC profile of the current kernel sources under profiles/"
    try:
        from fembrain_amd import lib as _fl
        rec = json.load(open(os.path.join(ROOT, "profiles", "r05_poly256_pmc.json")))
        if rec.get("kernel_source_sha256") == _fl.source_sha256("poly"):
            ftraffic = rec["kernels"]["k_tet_elements"]["bytes"]
            fnote = "profiles/r05_poly256_pmc.json (2 x FETCH_SIZE + WRITE_SIZE of k_tet_elements, separate --pmc passes; same kernel sources)"
        else:
            fnote = "profiles/r05_poly256_pmc.json was recorded for other kernel sources: not reported"
    except Exception:  # noqa: BLE001
        pass
    elem_bytes = 96.0 * n_inc + npts / 64.0 * 24.0
    # (round 5: the sweep stores f alone, 4 B per point; the float4 grid is materialised only for fb_poly_read_grid)
    pipe_bytes = 4.0 * npts + 12.0 * n_tv + 96.0 * n_inc + 3.0 * npts
    field_roofline = {"kernel": "k_tet_elements (6 tets of 16 B per included cell, one wavefront per run of 4 mask words, records transposed through LDS)",
                      "bound": "hbm", "achieved": elem_bytes / st[3] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": elem_bytes / st[3] / 1e9 / HBM_PEAK_GBS,
                      "algorithmic_bytes_per_launch": elem_bytes, "us_per_launch": st[3] * 1e6,
                      "traffic": ftraffic, "traffic_source": fnote,
                      "ceiling_measured": {"gbs": 6700.0, "what": "plain 16-byte store stream of this part, tools/ubench/writebw.hip (DESIGN.md section 4)"},
                      "stages_us": {"k_sweep": st[0] * 1e6, "classification_and_scans": st[1] * 1e6, "k_tet_vertices": st[2] * 1e6, "k_tet_elements": st[3] * 1e6,
                                    "all_with_events_between": st[4] * 1e6},
                      "pipeline": {"algorithmic_bytes": pipe_bytes, "us": pipe_s * 1e6, "achieved": pipe_bytes / pipe_s / 1e9, "frac": pipe_bytes / pipe_s / 1e9 / HBM_PEAK_GBS},
                      "sweep": {"what": "k_sweep: f alone, 4 B per point, + the inside mask (SURVEY 8d: 'store f only and say so'): compute-bound on the field, not a stream",
                                "algorithmic_bytes": 4.125 * npts, "us": sweep_s * 1e6, "achieved": 4.125 * npts / sweep_s / 1e9, "frac": 4.125 * npts / sweep_s / 1e9 / HBM_PEAK_GBS},
                      "sweep_and_xyzf_grid": {"what": "k_sweep + k_grid_xyzf: the float4 (x, y, z, f) grid fb_poly_read_grid / readBackVoxelGridSamples return, materialised on request "
                                                      "(16 B per point written, 4 read)", "algorithmic_bytes": 24.125 * npts, "us": grid_s * 1e6,
                                              "achieved": 24.125 * npts / grid_s / 1e9, "frac": 24.125 * npts / grid_s / 1e9 / HBM_PEAK_GBS,
                                              "mvoxels_per_s": npts / grid_s / 1e6}}
    out = {"field_roofline": field_roofline, "field_mvoxels_per_s": npts / pipe_s / 1e6, "field_sweep_mvoxels_per_s": npts / sweep_s / 1e6, "field_grid": list(dims),
           "field_sweep_gbs": npts * 4.125 / sweep_s / 1e9, "field_sweep_bytes_per_point": "4 (f) + 1 bit (inside); the 16-byte (x, y, z, f) grid on request: field_roofline.sweep_and_xyzf_grid",
           "field_pipeline_us": pipe_s * 1e6, "field_tets": int(p.counts.n_tets),
           "field_surface_us": surf_s * 1e6, "field_surface_vertices": int(p.counts.n_surface_vertices),
           "field_surface_triangles": int(p.counts.n_surface_indices) // 3,
           "field_pipeline": "sweep (f per point + inside mask) + edge/cell classification + scans + tet-mesh vertices and 6 tets per included cell"}
    if cpu:
        # bounded CPU sample: the oracle's scalar sweep + classification + tet emission on a 256x256x24 slab through the
        # sphere's equator (1 core), scaled to points/s; the reference CPU polygonizer is TBB-parallel over cores
        from oracle.pyfield import OrcPoly
        o = OrcPoly(blob)
        zs = 24
        t0 = time.perf_counter()
        o.sweep_grid((lower[0], lower[1], lower[2] + cell * 116), cell, (dims[0], dims[1], zs))
        o.classify()
        o.tetrahedralize()
        dt = time.perf_counter() - t0
        why_port = ("the reference's CPU polygonizer (src/implicit/Polygonizer.cpp) needs TBB headers this image lacks and its GPU path is OpenCL + GL: "
                    "neither builds here without stand-in headers, which the rules forbid; oracle/field_oracle.c is the scalar restatement pinned by the reference's shipped outputs")
        out["field_cpu_baseline"] = {"value": dims[0] * dims[1] * zs / dt / 1e6, "unit": "Mvoxels/s", "cores": 1, "kind": "port", "why_port": why_port,
                                     "sample": "oracle/field_oracle.c, 256x256x%d slab of the same grid through the sphere, sweep+classify+tets" % zs}
        # the reference's CPU polygonizer is TBB-parallel over cores (Polygonizer.cpp:627-629): the same slab on every core of the
        # box's share at once (threads; the C oracle runs outside the GIL), as a courtesy number
        from concurrent.futures import ThreadPoolExecutor
        cores = max(1, min(os.cpu_count() or 1, 16))

        def one(i):
            oo = OrcPoly(blob)
            oo.sweep_grid((lower[0], lower[1], lower[2] + cell * (40 + (i * 11) % 160)), cell, (dims[0], dims[1], zs))
            oo.classify()
            oo.tetrahedralize()
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(one, range(cores)))
        dta = time.perf_counter() - t0
        out["field_cpu_baseline_all_cores"] = {"value": cores * dims[0] * dims[1] * zs / dta / 1e6, "unit": "Mvoxels/s", "cores": cores, "kind": "port", "why_port": why_port,
                                               "sample": "%d threads, one 256x256x%d slab each" % (cores, zs)}
    p.close()
    return out


def field_bench_sharded(device, rank, world, allreduce):
    """The same 256^3 grid dealt to the ranks as z-slabs: every rank sweeps, classifies and tetrahedralizes its
    slab (one plane below, two above what it owns); the pieces are the single-GPU mesh bit for bit
    (tests/test_poly_gpu.py).  Rate = grid points / slowest rank's pipeline time; the only exchange is the all-gather of
    one vertex count.  allreduce(value, op) -> reduced float over the ranks."""
    from fembrain_amd.poly import GpuPoly, slab_plan, sphere_blob
    blob = sphere_blob()
    lower, cell, dims = (-0.5, -0.5, -0.5), 1.0 / 254.0, (256, 256, 256)
    p0, p1, z_first, z_count, own_planes, own_layers = slab_plan(dims[2], world, rank)
    p = GpuPoly(blob, device=device)
    p.sweep_slab(lower, cell, dims, z_first, z_count)
    p.classify()
    p.tetrahedralize()
    nv, nt = p.slab_counts(p0, own_planes, own_layers)
    sweep_s, pipe_s = p.time_pipeline(10)
    p.close()
    pipe = allreduce(pipe_s, "max")
    sweep = allreduce(sweep_s, "max")
    tets = int(round(allreduce(float(nt), "sum")))
    npts = dims[0] * dims[1] * dims[2]
    return {"field_mvoxels_per_s": npts / pipe / 1e6, "field_sweep_mvoxels_per_s": npts / sweep / 1e6, "field_grid": list(dims),
            "field_pipeline_us": pipe * 1e6, "field_tets": tets, "field_tets_expected": 39345600, "field_scaling": "strong",
            "field_partition": "z-slabs x%d, %d planes swept per rank for %d owned" % (world, z_count, own_planes)}


class BenchAbort(Exception):
    """a stage failed on some rank; every rank raises it together (agree()), rank 0 prints the line with an `error` field"""


_state = {"out": None, "rank": 0, "printed": False, "stage": "start-up", "soft_stage": None}
# iteration count of every persistent launch (one per solve) in launch order; FEMBRAIN_BENCH_LAUNCH_LOG=<path> writes it out for
# tools/summarize_profiles.py, which divides the per-launch PMC counters by it (bytes per PCG iteration)
_pipe_launch_iterations = []


def _emit(error=None):
    """rank 0's ONE JSON line: whatever has been measured so far, plus `error` when the run did not finish"""
    if _state["rank"] != 0 or _state["printed"]:
        return
    _state["printed"] = True
    out = dict(_state["out"] or {"metric": "FEM steps/sec (assemble+PCG) at 1M tets", "value": None, "unit": "steps/s"})
    if error is not None:
        out["error"] = error
    print(json.dumps(out), flush=True)
    log = os.environ.get("FEMBRAIN_BENCH_LAUNCH_LOG")
    if log:
        try:
            json.dump({"k_pcg_pipe_launch_iterations": _pipe_launch_iterations}, open(log, "w"))
        except OSError:
            pass


def _watchdog(seconds):
    """A rank that is stuck (a collective whose partner has left) cannot agree on anything: after `seconds` without a stage
    change it reports and leaves with a non-zero code, and the launcher tears the other ranks down.  Never restarts anything."""
    import threading

    def run():
        last, since = None, time.time()
        while True:
            time.sleep(5.0)
            if _state["printed"]:
                return
            if _state["stage"] != last:
                last, since = _state["stage"], time.time()
            elif _state.get("soft_stage") and time.time() - since > 90:
                # a leg that is only for the record (the collective-library leg, last of the run) did not come back: the line goes out
                # without it and the run counts -- every rank sits in the same stage and leaves the same way
                out = _state.get("out")
                if out is not None:
                    out.setdefault("config", {})["collective_library_leg"] = {"error": "no answer within 90 s in stage '%s'; everything else of the line was measured before" % last}
                _emit()
                os._exit(0)
            elif time.time() - since > seconds:
                _emit("stuck in stage '%s' for more than %d s on rank %d" % (last, seconds, _state["rank"]))
                os._exit(3)
    threading.Thread(target=run, daemon=True).start()


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) outside a launcher.  The parent never imports torch and never makes a HIP call (a process that has
    initialised the GPU must not be replaced or forked on this pool): it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>` as a child process, passes the child's stdout through line by
    line (rank 0's ONE JSON line among them), and returns the child's exit code.  If the ranks leave without a JSON line, the parent prints
    one with an `error` field, so that the caller always has a line to parse."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL and the peer-to-peer inboxes need it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    seen = False
    try:
        for line in child.stdout:
            if line.lstrip().startswith("{") and '"metric"' in line:
                seen = True
            sys.stdout.write(line)
            sys.stdout.flush()
        rc = child.wait()
    except KeyboardInterrupt:
        child.terminate()
        rc = child.wait()
    if os.environ.get("FEMBRAIN_BENCH_PARENT_TRACE") == "1":   # (tests: the parent stayed clear of torch / HIP)
        print("parent modules: torch=%s" % ("torch" in sys.modules), file=sys.stderr, flush=True)
    if not seen:
        print(json.dumps({"metric": "FEM steps/sec (assemble+PCG) at 1M tets", "value": None, "unit": "steps/s", "n_gpus": n,
                          "error": "the %d ranks started by bench.py left with code %d before rank 0 printed its line" % (n, rc),
                          "launched": " ".join(cmd)}), flush=True)
        rc = rc or 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cube56", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-field", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    _state["rank"] = rank
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` without a launcher: this process starts the N ranks itself (self_launch) -- as CHILD processes,
            # before torch or HIP has been touched here -- relays rank 0's line and leaves with the launcher's exit code
            sys.exit(self_launch(args.gpus))
        args.gpus = world

    import signal
    import torch
    from fembrain_amd import lib as fl
    from fembrain_amd.fem import FemIntegrator

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (the HIP path has no CPU fallback)")
    # The launcher ends the other ranks with SIGTERM when one leaves: rank 0 still prints what it has, with the reason.  The
    # main thread may sit inside a native call (a collective) where a Python-level handler never runs, so the signal is
    # picked up from the wake-up pipe by a helper thread.
    import threading
    rfd, wfd = os.pipe()
    os.set_blocking(wfd, False)
    signal.signal(signal.SIGTERM, lambda *_: None)
    signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)

    def on_term():
        while True:
            b = os.read(rfd, 1)
            if b and b[0] == signal.SIGTERM:
                _emit("terminated by the launcher in stage '%s' (a peer rank failed)" % _state["stage"])
                os._exit(4)
    threading.Thread(target=on_term, daemon=True).start()
    _watchdog(int(os.environ.get("FEMBRAIN_BENCH_STAGE_TIMEOUT_S", "420")))
    # FEMBRAIN_BENCH_LOCAL_COMM=1: rehearsal of the N > 1 flow on a ONE-GPU box -- every rank uses device 0, the process
    # group is gloo and the solver talks through the host-staged shared-memory communicator (RCCL refuses two ranks on
    # one device).  Numbers from this mode are meaningless; it exists to exercise the control flow.
    local_comm = os.environ.get("FEMBRAIN_BENCH_LOCAL_COMM") == "1"
    device = 0 if local_comm else local_rank
    torch.cuda.set_device(device)
    os.environ.setdefault("FEMBRAIN_P2P_TIMEOUT_MS", "5000")   # a healthy peer answers in microseconds; fail over to the collective library quickly
    shard = None
    comm = None
    # FEMBRAIN_BENCH_FORCE_DIST=1 runs the one-process-per-GPU plumbing (process group, unique-id broadcast, RCCL
    # communicator, sharded handle) even at world size 1 -- the rehearsal available on a one-GPU box
    dist_mode = world > 1 or os.environ.get("FEMBRAIN_BENCH_FORCE_DIST") == "1"
    dist = None
    if dist_mode:
        import ctypes as C
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        comm = C.c_void_p()
        pg_timeout = datetime.timedelta(seconds=300)   # a collective whose partner has left fails instead of waiting 10 minutes
        if local_comm:
            dist.init_process_group("gloo", timeout=pg_timeout)
            name = "/fembrain_bench_%s" % os.environ.get("MASTER_PORT", "0")
            fl.check(fl.lib().fb_comm_create_local(C.byref(comm), rank, world, name.encode(), 64 << 20, device))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device), timeout=pg_timeout)
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                buf = (C.c_ubyte * 128)()
                fl.check(fl.lib().fb_comm_unique_id(buf))
                uid = torch.tensor(list(buf), dtype=torch.uint8, device="cuda")
            dist.broadcast(uid, 0)
            idb = (C.c_ubyte * 128)(*uid.cpu().tolist())
            fl.check(fl.lib().fb_comm_create(C.byref(comm), rank, world, idb, device))
    tdev = "cpu" if local_comm else "cuda"
    comm_info = None
    if dist_mode:
        cr, cn, ck = C.c_int(0), C.c_int(0), C.c_int(0)
        fl.check(fl.lib().fb_comm_info(comm, C.byref(cr), C.byref(cn), C.byref(ck)))
        comm_info = {"ranks": cr.value, "rccl_ranks": cn.value, "transport": ["none", "RCCL (ncclCommInitRank over the launcher's unique id)",
                                                                              "host-staged shared-memory test transport (one-GPU rehearsal)"][ck.value]}

    def reduce_scalar(x, op):
        if not dist_mode:
            return float(x)
        tt = torch.tensor([x], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN, "sum": dist.ReduceOp.SUM}[op])
        return float(tt.item())

    def agree(ok):
        """did this succeed on EVERY rank?  (a peer-to-peer wait that times out returns FB_ECOMM after its bounded wait)"""
        return reduce_scalar(1.0 if ok else 0.0, "min") > 0.5

    def stage(name, fn, optional=False):
        """Runs fn on every rank, then lets the ranks agree on the outcome: either all carry on with their results, or all
        skip (optional stage: (None, reason)) / all abort (BenchAbort) TOGETHER -- no rank is left waiting in a barrier
        for a peer that took another path."""
        _state["stage"] = name
        err, res = None, None
        try:
            res = fn()
        except Exception as e:  # noqa: BLE001 -- whatever it is, the other ranks must learn about it
            err = "%s: %r" % (name, e)
        if agree(err is None):
            return res, None
        why = err or "%s: failed on another rank" % name
        if optional:
            return None, why
        raise BenchAbort(why)

    def barrier():
        if dist_mode:
            dist.barrier()
        torch.cuda.synchronize()

    rc = 0
    g = None
    try:
        n, text = WORKLOADS[args.workload]
        prec = fl.FB_MATRIX_F64 if args.precision == "f64" else fl.FB_MATRIX_F32

        # Sharded runs try the sharded persistent solver (one launch per solve on every rank, halo rows and rank sums crossing the
        # GPUs inside the launches; opt-in in the library because it is unmeasured on multi-GPU hardware): it is attached here, timed
        # as one more candidate beside the exchange modes, and checked like them before anything is timed.  Not in the one-GPU
        # rehearsal unless the ranks are confined to CU shares of their own (the persistent grids of two ranks must be resident together).
        if local_comm and os.environ.get("FEMBRAIN_BENCH_CU_SPLIT") == "1":   # rehearsal: every rank on its own share of the one GPU's CUs
            os.environ["FEMBRAIN_CU_MASK"] = "%d:%d" % (rank * (256 // world // 32 * 32), 256 // world // 32 * 32)
        if dist_mode and os.environ.get("FEMBRAIN_BENCH_NO_SHARDED_PERSIST") != "1" and (not local_comm or os.environ.get("FEMBRAIN_CU_MASK")):
            os.environ.setdefault("FEMBRAIN_SHARDED_PERSIST", "1")

        def create():
            v, t, fixed = workload_mesh(args.workload, device)
            sh = None
            if dist_mode:
                # slabs of whole i-planes (node index = i*n*n + j*n + k): <= 2 neighbours per rank
                if n:
                    planes = [n * r // world for r in range(world + 1)]
                    splits = np.array([p * n * n for p in planes], dtype=np.int32)
                else:  # grid-ordered polygonizer mesh: equal node ranges (z slabs of the voxel grid)
                    splits = np.array([len(v) * r // world for r in range(world + 1)], dtype=np.int32)
                sh = (world, rank, splits, comm)
            return v, t, fixed, sh, FemIntegrator(v, t, fixed, matrix_precision=prec, device=device, shard=sh)
        (v, t, fixed, shard, g), _ = stage("create the %s handle" % args.workload, create)

        def one_step(h=None):
            h = h or g
            h.rebuild_elements()
            h.set_uniform_force(1, -10000.0)
            it = h.do_timestep()
            if h.last.pcg_path == fl.FB_PCG_PATH_PERSISTENT:
                _pipe_launch_iterations.append(int(it))   # one persistent launch per solve: its iteration count, in launch order
            return it

        # Sharded runs: the exchange modes give bitwise identical iterates, so the fastest one ON THIS MACHINE is picked by
        # timing one step in each (max over ranks); untimed, before the warmup.  A form that fails on any rank ends the trials
        # for every rank alike and the collective library carries the run.
        xch_trials = {}
        xch_note = None
        sp_attached = dist_mode and g.sharded_persist()   # (agreed by the ranks when the handle was created)
        sp_use = False
        if sp_attached:
            def sp_trial():
                one_step()                                # first step (allocations, first-touch)
                barrier()
                ts = time.perf_counter()
                one_step()
                barrier()
                return time.perf_counter() - ts, int(g.last.pcg_path)
            sp_res, why = stage("exchange trial: sharded persistent launches", sp_trial, optional=True)
            ran = why is None and reduce_scalar(1.0 if sp_res[1] == fl.FB_PCG_PATH_PERSISTENT else 0.0, "min") > 0.5
            if ran:
                xch_trials["sharded_persistent"] = reduce_scalar(sp_res[0], "max") * 1e3
                sp_use = True
                g.set_sharded_persist(False)              # the exchange modes' turn
            else:
                xch_note = "the sharded persistent solver did not run its trial step (%s)" % (why or "a wait timed out, every rank fell back")
            g.reset_to_rest()
        if dist_mode and g.transport() >= fl.FB_XCH_P2P and os.environ.get("FEMBRAIN_XCH_MODE") is None:
            names = {fl.FB_XCH_COLLECTIVE: "collective", fl.FB_XCH_P2P: "p2p", fl.FB_XCH_P2P_SUMS: "p2p_sums", fl.FB_XCH_P2P_FUSED: "p2p_fused"}
            modes = (fl.FB_XCH_P2P, fl.FB_XCH_P2P_SUMS, fl.FB_XCH_P2P_FUSED)
            _, why = stage("exchange trial: first step", one_step, optional=True)
            for mode in modes if why is None else ():
                def trial():
                    g.set_exchange_mode(mode)
                    barrier()
                    ts = time.perf_counter()
                    one_step()
                    barrier()
                    return time.perf_counter() - ts
                dt_trial, why = stage("exchange trial: %s" % names[mode], trial, optional=True)
                if why is not None:
                    break
                xch_trials[names[mode]] = reduce_scalar(dt_trial, "max") * 1e3
            if why is None:
                best = min((k for k in xch_trials if k != "sharded_persistent"), key=xch_trials.get)
                g.set_exchange_mode({vv: k for k, vv in names.items()}[best])
                sp_use = sp_use and xch_trials["sharded_persistent"] <= xch_trials[best]
            else:
                # a peer-to-peer form failed on some rank (its inbox is poisoned from then on): the collective library carries
                # the rest of the run, on every rank alike, and the line says so
                xch_note = "peer-to-peer exchange failed in the trial step (%s); fell back to the collective library" % why
                g.set_exchange_mode(fl.FB_XCH_COLLECTIVE)
            g.reset_to_rest()   # the timed steps start from the same state as the one-GPU run's
        if sp_attached and g.persist_info()[0] != sp_use and not (sp_use and g.pcg_path()["fallbacks"]):
            g.set_sharded_persist(sp_use)
        # Sharded runs check themselves before anything is timed: the first step from rest, gathered over the ranks, against
        # the same step of an UNSHARDED handle on rank 0's GPU (iteration count within max(3, 2 %), displacements within 2e-4
        # of max|q| -- the tolerance of the parity tests for two solves that both stop at a 1e-6 residual; the same check runs
        # under pytest at 8M tets: tests/test_sharded_gpu.py).  If a peer-to-peer form fails the check, the collective library
        # takes over and is checked the same way; if that fails too the run aborts: a wrong solver is not timed.
        sharded_check = None
        if dist_mode and os.environ.get("FEMBRAIN_BENCH_NO_CHECK") != "1":
            def reference_step():
                if rank != 0:
                    return None
                g1 = FemIntegrator(v, t, fixed, matrix_precision=prec, device=device)
                it_ref = one_step(g1)
                q_ref = g1.get_q_state()[0]
                g1.close()
                return it_ref, q_ref
            ref, _ = stage("self-check: unsharded reference step on rank 0", reference_step)

            def check_once():
                g.reset_to_rest()
                its, why = stage("self-check: sharded step", one_step, optional=True)
                q = g.get_q_state()[0] if why is None else np.zeros(g.r)
                tq = torch.from_numpy(q.copy()).to(tdev)
                dist.all_reduce(tq, op=dist.ReduceOp.SUM)   # every rank fills its owned range only
                res = {"ok": False, "error": why} if why is not None else None
                good = 0.0
                if rank == 0 and why is None:
                    qa = tq.cpu().numpy()
                    diff = float(np.abs(qa - ref[1]).max() / np.abs(ref[1]).max())
                    res = {"iterations_sharded": int(its), "iterations_one_gpu": int(ref[0]), "max_rel_diff_q": diff,
                           "ok": bool(abs(its - ref[0]) <= max(3, 0.02 * ref[0]) and diff <= 2e-4)}
                    good = 1.0 if res["ok"] else 0.0
                return reduce_scalar(good, "max") > 0.5, res

            passed, sharded_check = check_once()
            if not passed and sp_attached and g.sharded_persist():
                first = sharded_check
                xch_note = ((xch_note + "; ") if xch_note else "") + "the sharded persistent solver failed the self-check, the two-launch iteration took over"
                g.set_sharded_persist(False)
                passed, sharded_check = check_once()
                if rank == 0 and sharded_check is not None:
                    sharded_check["sharded_persistent_attempt"] = first
            if not passed and g.transport() >= fl.FB_XCH_P2P:
                first = sharded_check
                xch_note = ((xch_note + "; ") if xch_note else "") + "the peer-to-peer exchange failed the self-check, the collective library took over"
                g.set_exchange_mode(fl.FB_XCH_COLLECTIVE)
                passed, sharded_check = check_once()
                if rank == 0 and sharded_check is not None:
                    sharded_check["peer_to_peer_attempt"] = first
            if not passed:
                _state["out"] = {"metric": "FEM steps/sec (assemble+PCG) at 1M tets", "value": None, "unit": "steps/s", "n_gpus": world,
                                 "config": {"sharded_self_check": sharded_check, "exchange_note": xch_note}}
                raise BenchAbort("the sharded solver failed its self-check against the unsharded handle")
            g.reset_to_rest()

        def timed():
            for _ in range(args.warmup):
                one_step()
            barrier()
            t0 = time.perf_counter()
            iters, asm_s, solve_s = [], 0.0, 0.0
            for _ in range(args.steps):
                iters.append(one_step())
                asm_s += g.last.assembly_seconds
                solve_s += g.last.solve_seconds
            barrier()
            return time.perf_counter() - t0, iters, asm_s, solve_s
        (dt, iters, asm_s, solve_s), _ = stage("warm-up and timed steps", timed)
        dt = reduce_scalar(dt, "max")

        # The trajectory above continues a loaded dynamic simulation, so its iteration counts depend on --warmup / --steps.
        # Beside it: the same step from the SAME state every time (reset to rest before each timed step; only the step is timed).
        def fixed_state():
            tot, its = 0.0, []
            for _ in range(args.steps):
                g.reset_to_rest()
                barrier()
                ts = time.perf_counter()
                its.append(one_step())
                barrier()
                tot += time.perf_counter() - ts
            return tot, its
        fs, fs_why = stage("steps from the rest state", fixed_state, optional=True)
        fixed_dt, fixed_iters = (reduce_scalar(fs[0], "max"), fs[1]) if fs else (None, None)

        def probes():
            # dominant kernel: the PCG SpMV.  Average launch duration measured with HIP events on the handle's stream.
            spmv_s = g.time_spmv(200)
            persist = g.persist_info()
            persist_s = g.persist_stats() if persist[0] else None   # every persistent launch this handle's solves have made: (launches, device seconds, iterations)
            asm_k_s = g.time_assembly(10)
            halo_s, sum_s = g.time_exchange(200) if dist_mode else (0.0, 0.0)
            k0_s = g.time_element_stiffness(3)
            resync_ms = None
            if not dist_mode:  # Deformable::syncForceModel after a cut: plan (pattern, SELL, contribution lists) + rest state
                ts = time.perf_counter()
                g.resync(v, t, fixed)
                torch.cuda.synchronize()
                resync_ms = (time.perf_counter() - ts) * 1e3
            return spmv_s, g.spmv_bytes(), asm_k_s, halo_s, sum_s, k0_s, resync_ms, persist, persist_s, g.iteration_bytes()
        (spmv_s, spmv_bytes, asm_k_s, halo_s, sum_s, k0_s, resync_ms, persist, persist_s, iter_bytes), _ = stage("kernel probes", probes)
        slots_total = int(fl.lib().fb_fem_device_plan_get(g.h, b"slot_coff", None, 0))   # SELL slots (64 block rows each) of this rank's matrix

        if rank == 0:
            # HBM traffic of the dominant kernel from the PMC counters: taken from the committed profile only while the kernel
            # sources are still the ones that were profiled (tools/summarize_profiles.py records their hash), else null
            traffic, traffic_note = None, "no PMC profile of the current kernel sources under profiles/"
            dominant = "k_pcg_pipe" if persist[0] else "k_spmv"
            pmc = os.path.join(ROOT, "profiles", "dominant_pmc.json")
            if os.path.exists(pmc) and args.workload == "cube56" and world == 1 and args.precision == "f32":
                try:
                    rec = json.load(open(pmc))
                    if rec.get("kernel_source_sha256") != fl.source_sha256("fem"):
                        traffic_note = "profiles/dominant_pmc.json was recorded for other kernel sources: not reported"
                    elif not rec.get("kernel", "").startswith(dominant):
                        traffic_note = "profiles/dominant_pmc.json is for %s, this run's dominant kernel is %s" % (rec.get("kernel"), dominant)
                    else:
                        traffic, traffic_note = rec.get("hbm_bytes_per_unit" if persist[0] else "hbm_bytes_per_launch"), "profiles/dominant_pmc.json (same kernel sources)"
                except Exception:
                    pass
            spmv_roof = {"kernel": "k_spmv (SELL-64 3x3-block SpMV of the two-launch PCG iteration; the exact-residual iterations and systems "
                                   "outside the persistent kernel's range run it)",
                         "achieved": spmv_bytes / spmv_s / 1e9, "frac": spmv_bytes / spmv_s / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": spmv_bytes, "us_per_launch": spmv_s * 1e6,
                         # for comparison, SURVEY 8d's plain BSR figure (4-byte column ids, x read once, y written once, here in fp64)
                         "survey_bsr_bytes_per_launch": (g.num_blocks() * 40.0 + (len(v) + 1) * 4.0 + 3.0 * len(v) * 16.0) if shard is None else None}
            if persist[0]:
                # dominant kernel: k_pcg_pipe, ONE launch = ONE whole solve (every iteration of a step).  Unit = one PCG iteration;
                # algorithmic bytes per unit = SURVEY 8(d): BSR SpMV + the fused lower bound of the vector traffic
                # (fb_fem_iteration_bytes); the exact-residual products every 30th iteration are NOT counted as work.  Launch
                # durations: HIP events on the handle's stream around EVERY persistent launch this process made (warm-up, timed
                # and fixed-state steps alike), so the average is the one a kernel trace of the same command shows.
                # The kernel keeps the vectors and part of the matrix on-chip, so its HBM/L3 traffic (PMC) is BELOW the
                # algorithmic figure -- the opposite of wasted re-reads; bus_frac = that traffic / time / peak.
                n_launch, sec, n_it = persist_s
                upl = n_it / max(n_launch, 1)
                work = n_it * iter_bytes / sec / 1e9          # work-equivalent rate: algorithmic bytes / time (exceeds the peak: see below)
                # Bytes that really cross the L2 <-> fabric boundary per iteration.  Measured: PMC 2 x FETCH_SIZE + WRITE_SIZE from the
                # committed profile of these kernel sources.  Without a matching profile: a MODEL of the same quantity -- the streamed
                # slots of the matrix (values + column words), the published vector written through once and fetched once per XCD that
                # gathers it (~1.5x), flags and sums -- labelled as such.
                streamed = max(0.0, 1.0 - persist[3] * persist[1] * persist[2] / max(1.0, float(slots_total))) if slots_total else 1.0
                model = slots_total * 64.0 * (36.0 + (2.0 if "c16" in g.pcg_path()["kernel"] else 4.0)) * streamed + 2.5 * 24.0 * len(v)
                bus_unit, bus_src = (traffic, traffic_note) if traffic is not None else (model, "MODEL (no PMC profile of the current kernel sources under profiles/): "
                                                                                         "streamed matrix slots + published vector stores and fetches")
                bus = bus_unit * n_it / sec / 1e9
                roofline = {"kernel": "%s (pipelined Jacobi-PCG, one launch per solve: product + sums + recurrences of every iteration, vectors in "
                                      "registers, %d of ~15 slots of every slice resident in LDS)" % (g.pcg_path()["kernel"], persist[3]),
                            "bound": "hbm", "achieved": bus, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bus / HBM_PEAK_GBS,
                            "frac_is": "bytes that cross the L2 <-> fabric boundary (PMC 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction) / time / 8 TB/s: what the "
                                       "memory system really sustains.  The algorithmic bytes of the same iterations (SURVEY 8d: BSR SpMV + 9 vector "
                                       "streams) are under work_equivalent_*; their rate exceeds the peak because two fifths of them never leave the "
                                       "registers / LDS",
                            "traffic": bus_unit * upl, "traffic_per_unit": bus_unit, "traffic_source": bus_src,
                            "traffic_measured": traffic is not None,
                            "work_equivalent_gbs": work, "work_equivalent_frac": work / HBM_PEAK_GBS,
                            "served_from": "the 256 MiB Infinity Cache: the system of an iteration (100 MB of matrix, 12 MB of vectors) stays resident between "
                                           "iterations, so HBM itself is nearly idle; FETCH_SIZE / WRITE_SIZE count L2-to-fabric requests, cache hits included",
                            "ceiling_measured": {"gbs": 6600.0, "what": "streaming read of a 102 MB Infinity-Cache-resident buffer on this part "
                                                                        "(tools/ubench/readbw.hip; 6.2 TB/s for 1 GB from HBM)",
                                                 "frac_of_ceiling": bus / 6600.0},
                            "launches": n_launch, "units_per_launch": upl, "algorithmic_bytes_per_unit": iter_bytes,
                            "algorithmic_bytes_per_launch": upl * iter_bytes, "us_per_launch": sec / n_launch * 1e6, "us_per_unit": sec / n_it * 1e6,
                            "wavefronts_per_cu": persist[1], "workgroups": persist[2], "max_producers_per_workgroup": g.pcg_path()["max_producers"],
                            "persist_fallbacks": g.pcg_path()["fallbacks"], "persist_rearms": int(fl.lib().fb_fem_persist_rearms(g.h)), "spmv_kernel": spmv_roof}
            else:
                roofline = dict(spmv_roof, bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s", traffic=traffic, traffic_source=traffic_note)
            storage = "f64 arithmetic / f32 stored matrix" if args.precision == "f32" else "f64"
            _state["out"] = {
                "metric": "FEM steps/sec (assemble+PCG) at 1M tets" if args.workload == "cube56" else "FEM steps/sec (assemble+PCG)", "value": args.steps / dt, "unit": "steps/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                # the arithmetic type of the path: every product, sum and vector is fp64; only the STORED matrix values are fp32 by
                # default (the reference stores fp64: sparseMatrix.h:352-356; the north star allows fp32 within a stated tolerance)
                "dtype": storage, "data": "synthetic",
                "config": {"workload": text, "nodes": int(len(v)), "tets": int(len(t)), "partition": "i-plane slabs x%d" % world, "matrix_storage": args.precision,
                           "exchange": ["none (one GPU)", "host-staged test communicator (rehearsal)" if local_comm else "RCCL all-reduce + send/recv",
                                        "peer-to-peer inboxes over xGMI (HIP IPC), one kernel per exchange",
                                        "peer-to-peer inboxes, sums inside the PCG kernels",
                                        "peer-to-peer inboxes, sums and halo values inside the PCG kernels"][g.transport()]
                                       if not (dist_mode and g.sharded_persist()) else
                                       "sharded persistent launches: halo rows stored into the neighbour rank's box and rank sums posted to every rank "
                                       "inside the one launch per solve (HIP IPC over xGMI, no exchange kernel, no collective)",
                           "exchange_trials_ms_per_step": xch_trials, "exchange_note": xch_note, "sharded_self_check": sharded_check,
                           "cg_eps": 1e-6, "cg_max_iter": 10000,
                           "pcg": ("one persistent launch per solve, pipelined iteration (%s)" % ("FB_PCG_PERSISTENT, default at this size" if shard is None
                                                                                                    else "k_pcg_pipe_shard, on every rank") if persist[0]
                                   else "two launches per merged iteration (FB_PCG_MERGED)"),
                           "pcg_path_last_step": int(g.last.pcg_path), "persist_fallbacks": int(g.last.persist_fallbacks)},
                "cg_iterations": [int(i) for i in iters], "cg_iterations_per_step": float(np.mean(iters)),
                "value_note": "timed steps continue the loaded simulation after the warm-up steps (trajectory); value_at_fixed_state times the same "
                              "step from the rest state every time",
                "value_at_fixed_state": (args.steps / fixed_dt) if fixed_dt else None,
                "cg_iterations_at_fixed_state": [int(i) for i in fixed_iters] if fixed_iters else fs_why,
                "assembly_ms_per_step": asm_s / args.steps * 1e3,
                "solve_ms_per_step": solve_s / args.steps * 1e3, "us_per_cg_iteration": solve_s / max(sum(iters), 1) * 1e6,
                "resync_ms": resync_ms,
                # informational (north star: MFMA only for the batched 12x12 element contractions): forming every K0 = V B^T E B on the
                # fp64 matrix cores; 2*(6*6*12 + 12*6*12) flop and 1152 B written per element.  The per-step path never forms K0.
                "element_k0_mfma": {"on_step_path": False,
                                    "note": "not part of a step: the step path never forms K0 (block (i,j) of R K0 R^T is closed-form from a 64-byte "
                                            "record per tet, DESIGN.md section 3); this kernel serves fb_fem_element_stiffness (inspection) only",
                                    "us_per_pass": k0_s * 1e6, "gflops": 2592.0 * len(t) / k0_s / 1e9 if shard is None else None,
                                    "write_gbs": 1152.0 * len(t) / k0_s / 1e9 if shard is None else None},
                "exchange_us": {"halo_refresh": halo_s * 1e6, "global_sum_3": sum_s * 1e6},
                "assembly_kernels_us": asm_k_s * 1e6, "assembly_gbs": g.assembly_bytes() / asm_k_s / 1e9,
                "roofline": roofline,
                "cpu_baseline": None,
            }
        out = _state["out"]
        mode_used = g.transport()
        sp_final = bool(dist_mode and g.sharded_persist())
        if out is not None:
            out["config"]["communicator"] = comm_info
            out["config"]["rccl_ranks"] = comm_info["rccl_ranks"] if comm_info else 0
        if not dist_mode:
            g.close()
            g = None
        # (a sharded run keeps its handle for the collective-library leg, which runs LAST: after the field and the 8M-tet legs)

        # ---- optional legs: each is entered and left by all ranks together; a failure is recorded, the headline survives ----
        if dist_mode and world > 1 and not args.no_field:
            fs, why = stage("field grid dealt to the ranks", lambda: field_bench_sharded(device, rank, world, reduce_scalar), optional=True)
            if out is not None:
                out.update(fs if fs else {"field_error": why})
        if world == 1 and not args.no_field:
            extra, why = stage("field bench", lambda: field_bench(device, cpu=not args.no_cpu_baseline), optional=True)
            if out is not None:
                out.update(extra if extra else {"field_error": why})
        # BASELINE configs 1-2 and the numbering leg, one GPU only (driver-timed, never part of `value`)
        if world == 1 and args.workload == "cube56" and os.environ.get("FEMBRAIN_BENCH_SKIP_LEGS") != "1":
            for key, wl in (("cube27", "cube27"), ("blob100k", "ventricle")):
                leg, why = stage("leg: %s" % key, lambda wl=wl: small_leg(wl, device, prec, cpu=not args.no_cpu_baseline), optional=True)
                if out is not None:
                    out[key] = leg if leg else {"error": why}
            leg, why = stage("leg: scrambled node order", lambda: numbering_leg(device, prec), optional=True)
            if out is not None:
                out["cube56_scrambled"] = leg if leg else {"error": why}
            leg, why = stage("leg: unstructured mesh", lambda: unstructured_leg(device, prec), optional=True)
            if out is not None:
                out["delaunay606k"] = leg if leg else {"error": why}
            leg, why = stage("leg: re-sync after a cut", lambda: cut_resync_leg(device, prec), optional=True)
            if out is not None:
                out["cut_resync"] = leg if leg else {"error": why}
            leg, why = stage("leg: opt-in block-Jacobi", lambda: block_jacobi_leg(device, prec), optional=True)
            if out is not None:
                out["cube56_block_jacobi_opt_in"] = leg if leg else {"error": why}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            _state["stage"] = "cpu baseline"
            try:
                out["cpu_baseline"] = cpu_baseline((v, t, fixed), int(round(np.mean(iters))))
                out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline_error"] = repr(e)
        # BASELINE config 5 in the same run: the 8M-tet cube (111^3 nodes) on the same ranks, 1 warm-up + 2 timed steps, so that
        # the N = 1, 2, 4, 8 runs of this script also give the 8M-tet strong-scaling series of the north star.  Reported
        # under "cube111"; never part of `value`.  Every stage of the leg is agreed on by all ranks.
        if args.workload == "cube56" and os.environ.get("FEMBRAIN_BENCH_SKIP_8M") != "1":
            big, g8 = None, None

            steps8 = max(2, int(os.environ.get("FEMBRAIN_BENCH_8M_STEPS", "5")))
            big_name = os.environ.get("FEMBRAIN_BENCH_BIG_WORKLOAD", "cube111")   # (rehearsals on one GPU: a mesh whose shards the sharded persistent solver takes)

            def create8():
                n8 = WORKLOADS[big_name][0]
                v8, t8, fixed8 = workload_mesh(big_name, device)
                shard8 = None
                if dist_mode:
                    planes = [n8 * r // world for r in range(world + 1)]
                    shard8 = (world, rank, np.array([p * n8 * n8 for p in planes], dtype=np.int32), comm)
                h8 = FemIntegrator(v8, t8, fixed8, matrix_precision=prec, device=device, shard=shard8)
                return h8, len(t8), (shard8[2] if shard8 else None)
            made, why = stage("8M-tet leg: create", create8, optional=True)
            if made:
                g8, ntets8, shard8_splits = made
                if dist_mode and mode_used != g8.transport() and g8.transport() >= fl.FB_XCH_P2P:
                    g8.set_exchange_mode(mode_used)   # collective; the form picked (or fallen back to) above
                sp8 = None
                if dist_mode and g8.sharded_persist():
                    # The sharded persistent solver is decided HERE for this mesh (at N = 8 a rank holds 1M of the 8M tets -- its
                    # best case -- while the 1M-tet mesh above left it 125k): the first step from rest with it and with the
                    # two-launch iteration; it runs the timed steps if the two agree (iteration counts within max(3, 2 %), every
                    # rank's owned displacements within 2e-4 of max|q|) and it was the faster one on the slowest rank.
                    def trial8():
                        res = {}
                        for on in (True, False):
                            g8.set_sharded_persist(on)
                            g8.reset_to_rest()
                            barrier()
                            ts = time.perf_counter()
                            it = one_step(g8)
                            barrier()
                            res[on] = (time.perf_counter() - ts, int(it), g8.get_q_state()[0], int(g8.last.pcg_path))
                        return res
                    r8, why8 = stage("8M-tet leg: sharded persistent trial", trial8, optional=True)
                    if r8 is not None:
                        lo8, hi8 = 3 * int(shard8_splits[rank]), 3 * int(shard8_splits[rank + 1])
                        dq = float(np.abs(r8[True][2][lo8:hi8] - r8[False][2][lo8:hi8]).max())
                        qm = float(np.abs(r8[False][2][lo8:hi8]).max())
                        dq, qm = reduce_scalar(dq, "max"), reduce_scalar(qm, "max")
                        ran = reduce_scalar(1.0 if r8[True][3] == fl.FB_PCG_PATH_PERSISTENT else 0.0, "min") > 0.5
                        t_on, t_off = reduce_scalar(r8[True][0], "max"), reduce_scalar(r8[False][0], "max")
                        agree8 = ran and abs(r8[True][1] - r8[False][1]) <= max(3, 0.02 * r8[False][1]) and dq <= 2e-4 * max(qm, 1e-300)
                        use8 = bool(agree8 and t_on <= t_off)
                        sp8 = {"ms_step_sharded_persistent": t_on * 1e3, "ms_step_two_launch": t_off * 1e3, "iterations": [r8[True][1], r8[False][1]],
                               "max_rel_diff_q": dq / max(qm, 1e-300), "ran_persistent": bool(ran), "used": use8}
                        if use8 and not g8.pcg_path()["fallbacks"]:
                            stage("8M-tet leg: sharded persistent on", lambda: g8.set_sharded_persist(True), optional=True)
                    else:
                        sp8 = {"error": why8, "used": False}
                        stage("8M-tet leg: sharded persistent off", lambda: g8.set_sharded_persist(False), optional=True)
                    g8.reset_to_rest()
                _, why = stage("8M-tet leg: warm-up step", lambda: one_step(g8), optional=True)
            if made and why is None:
                def timed8():
                    barrier()
                    ts = time.perf_counter()
                    it8, solve8 = [], 0.0
                    for k8 in range(steps8):
                        if os.environ.get("FEMBRAIN_BENCH_INJECT_FAILURE") == "8m" and rank == world - 1 and k8 == 1:
                            raise RuntimeError("injected failure (rehearsal of the failure path)")
                        it8.append(one_step(g8))
                        solve8 += g8.last.solve_seconds
                    barrier()
                    return time.perf_counter() - ts, it8, solve8
                res8, why = stage("8M-tet leg: timed steps", timed8, optional=True)
                if res8:
                    dt8 = reduce_scalar(res8[0], "max")
                    it8, solve8 = res8[1], res8[2]
                    spmv8 = None
                    if not dist_mode:
                        try:
                            spmv8 = g8.spmv_bytes() / g8.time_spmv(50) / 1e9
                        except Exception:  # noqa: BLE001
                            pass
                    us_it_rank = solve8 / max(sum(it8), 1) * 1e6
                    if dist_mode:   # every rank's own figure (solve seconds are device time on the rank's stream)
                        tl = torch.zeros(world, dtype=torch.float64, device=tdev)
                        tl[rank] = us_it_rank
                        dist.all_reduce(tl, op=dist.ReduceOp.SUM)
                        us_it_ranks = [float(x) for x in tl.cpu().tolist()]
                    else:
                        us_it_ranks = [us_it_rank]
                    big = {"workload": WORKLOADS[big_name][1], "tets": int(ntets8), "steps": steps8, "warmup": 1, "value": steps8 / dt8, "unit": "steps/s",
                           "ms_per_step": dt8 / steps8 * 1e3, "us_per_cg_iteration_per_rank": us_it_ranks, "cg_iterations": [int(i) for i in it8], "cg_iterations_per_step": float(np.mean(it8)),
                           "us_per_cg_iteration": solve8 / max(sum(it8), 1) * 1e6, "spmv_gbs": spmv8,
                           "pcg_kernel": (g8.pcg_path()["kernel"] if g8.last.pcg_path == fl.FB_PCG_PATH_PERSISTENT else "") or "two-launch iteration",
                           "pcg_path_last_step": int(g8.last.pcg_path), "sharded_persistent_trial": sp8}
            if g8 is not None:
                g8.close()
            if out is not None:
                out["cube111"] = big if big else {"error": why}
                # the north star's multi-GPU question in one place: the 8M-tet mesh on these N ranks against the
                # committed one-GPU figure of the same leg (profiles/n1_8m_reference.json, written from a --gpus 1 run of this script)
                if big:
                    ref8 = None
                    try:
                        ref8 = json.load(open(os.path.join(ROOT, "profiles", "n1_8m_reference.json")))
                    except Exception:  # noqa: BLE001
                        pass
                    same = bool(ref8) and ref8.get("tets") == big["tets"]
                    out["scaling_8m"] = {"workload": big["workload"], "n_gpus": world, "steps_per_s": big["value"], "ms_per_step": big["ms_per_step"],
                                         "us_per_cg_iteration_per_rank": big["us_per_cg_iteration_per_rank"], "cg_iterations_per_step": big["cg_iterations_per_step"],
                                         "pcg_kernel": big["pcg_kernel"], "scaling": "strong (the same 8M-tet mesh on every N)",
                                         "n1_steps_per_s_committed": ref8.get("steps_per_s") if same else None,
                                         "n1_source": ref8.get("source") if same else "profiles/n1_8m_reference.json holds no figure for this mesh",
                                         "speedup_over_n1": (big["value"] / ref8["steps_per_s"]) if same and ref8.get("steps_per_s") else None,
                                         "target": "north star: >= 6x at N = 8"}
        # The north star's transport for the record: ONE step from rest with every exchange of every PCG iteration going
        # through the collective library (RCCL all-reduce of the three sums + ncclSend/ncclRecv of the halo rows; FB_XCH_COLLECTIVE), and the
        # same step in the form the timed steps used.  LAST of all legs (the 8M-tet scaling leg included) and under a watchdog of its own (90 s): a collective
        # that never returns costs this leg -- the line goes out without it, exit code 0 -- not the run.  FEMBRAIN_BENCH_TRY_RCCL=0 skips it.
        if dist_mode and g is not None and os.environ.get("FEMBRAIN_BENCH_TRY_RCCL", "1") != "0":
            _state["soft_stage"] = "collective-library leg"
            def rest_step(mode, sp):
                if sp_attached and g.persist_info()[0] != sp and not (sp and g.pcg_path()["fallbacks"]):
                    g.set_sharded_persist(sp)
                if g.transport() != mode:
                    g.set_exchange_mode(mode)
                g.reset_to_rest()
                barrier()
                ts = time.perf_counter()
                it = one_step()
                barrier()
                return time.perf_counter() - ts, int(it), g.last.solve_seconds
            leg = {}
            for key, mode, sp in (("collective", fl.FB_XCH_COLLECTIVE, False), ("as_timed", mode_used, sp_final)):
                res, why = stage("collective-library leg: %s" % key, lambda mode=mode, sp=sp: rest_step(mode, sp), optional=True)
                if res is None:
                    leg[key] = {"error": why}
                    break
                ms = reduce_scalar(res[0], "max") * 1e3
                leg[key] = {"ms_per_step": ms, "cg_iterations": res[1], "us_per_cg_iteration": reduce_scalar(res[2], "max") / max(res[1], 1) * 1e6}
            if out is not None:
                if "ms_per_step" in leg.get("collective", {}):
                    out["config"]["exchange_trials_ms_per_step"] = dict(out["config"]["exchange_trials_ms_per_step"], collective=leg["collective"]["ms_per_step"])
                out["config"]["collective_library_leg"] = dict(leg, what="one step from the rest state on the same handle: every exchange through "
                                                               "the collective library (FB_XCH_COLLECTIVE) against the form the timed steps used",
                                                               communicator=comm_info)
            _state["soft_stage"] = None
        if g is not None:
            g.close()
            g = None
    except BenchAbort as e:
        rc = 1
        _emit(str(e))
    except Exception as e:  # noqa: BLE001 -- outside any agreed stage: report and leave non-zero, the launcher ends the peers
        rc = 2
        _emit("%s (stage '%s', rank %d)" % (repr(e), _state["stage"], rank))
    if rc == 0:
        _state["stage"] = "shutdown"
        if g is not None:
            g.close()
        if dist_mode:
            fl.lib().fb_comm_destroy(comm)
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001 -- nothing is measured after this point
                pass
        _emit()
    sys.stdout.flush()
    if rc != 0:
        os._exit(rc)   # do not wait in destructors for peers that may be gone


if __name__ == "__main__":
    main()
