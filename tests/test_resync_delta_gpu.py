"""fb_fem_resync_delta (fembrain_amd/csrc/delta.h): the re-sync after a cut from a DESCRIPTION of the change.

CuttableMesh::cut erases the cut cells keeping the order of the rest (src/deformable/VolMesh.cpp:630), appends their pieces and the new
nodes (:1083-1088) and re-points cells in place (:1630-1650); Deformable::syncForceModel (src/deformable/Deformable.cpp:127-220) then
rebuilds everything.  The handle keeps its mesh on the device and updates the sorted pair list its plan was built from.  Checked here
against a handle made from the whole new mesh: every plan array, the assembled matrix and force, and a step, bit for bit where the
handle works in the caller's node order; pattern, values and the step to rounding where the handle has an internal order of its own.
The change is meshgen.synthetic_cut (every element crossing a plane split in four on a new centroid node: removed, changed in place,
appended) -- not the reference's subdivision tables, which live in the host's CuttableMesh.
"""
import numpy as np
import pytest

from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator, bsr_to_scipy
from fembrain_amd.meshgen import apply_delta, cube_fixed_plane_i0, fixed_vertices_to_dofs, synthetic_cut, truth_cube

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _keep_the_merged_path_busy(monkeypatch):
    """By default a renumbered handle asks for a fresh node order once 2 % more nodes have come than its order was built for (a fresh order
    pays for its rebuild within one step, fem.hip kFreshOrderPercent).  The tests below exercise the MERGED path with larger changes on
    small meshes, so they run with the limit at a tenth (read at every call); test_fresh_order_rule checks the default."""
    monkeypatch.setenv("FEMBRAIN_FRESH_ORDER_PERCENT", "10")


PLAN = ("bptr", "bcol", "blk_slot", "slice_off", "colidx", "slot_coff", "slot_ccnt", "contrib")


def _cube(n):
    v, t = truth_cube(n, n, n, 0.1)
    return v, t, fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))


def _device_plan(g, name):
    L = fl.lib()
    n = L.fb_fem_device_plan_get(g.h, name.encode(), None, 0)
    assert n >= 0, (name, L.fb_last_error())
    a = np.zeros(n, np.int32)
    assert L.fb_fem_device_plan_get(g.h, name.encode(), fl.iptr(a), n) == n
    return a


def _same_bits(ga, gb, seed=5, load=-2000.0):
    """plan arrays, assembled force and matrix, and two steps of two handles: bit for bit"""
    for name in PLAN:
        assert np.array_equal(_device_plan(ga, name), _device_plan(gb, name)), name
    assert ga.r == gb.r
    u = np.random.default_rng(seed).normal(size=ga.r) * 0.003
    fa, Ka = ga.assemble(u)
    fb_, Kb = gb.assemble(u)
    assert np.array_equal(fa, fb_) and np.array_equal(Ka, Kb)
    assert np.array_equal(ga.mass(), gb.mass())
    for g in (ga, gb):
        g.reset_to_rest()
    for k in range(2):
        its = []
        for g in (ga, gb):
            g.set_uniform_force(1, load)
            its.append(g.do_timestep())
        assert its[0] == its[1] > 0
        assert np.array_equal(ga.get_q_state()[0], gb.get_q_state()[0])


@pytest.mark.parametrize("n,axis,every_changed", [(10, 0, 3), (14, 1, 3), (14, 2, 0), (12, 1, 1)])
def test_delta_resync_gives_the_plan_and_the_step_of_a_full_resync_bit_for_bit(gpu, n, axis, every_changed):
    v, t, fixed = _cube(n)
    g = FemIntegrator(v, t, fixed)
    g.set_uniform_force(1, -500.0)
    g.do_timestep()                     # (a state to be reset)
    v2, t2, d = synthetic_cut(v, t, axis=axis, where=0.4, every_changed=every_changed)
    assert len(d["removed"]) + len(d["changed_ids"]) > 0 and len(d["added"]) > 0 and len(d["new_xyz"]) > 0
    g.resync_delta(d, fixed)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED
    assert fl.lib().fb_fem_num_nodes(g.h) == len(v2) and fl.lib().fb_fem_num_tets(g.h) == len(t2)
    assert not np.any(g.get_q_state()[0])
    ref = FemIntegrator(v2, t2, fixed)
    assert ref.resync_path() == fl.FB_RESYNC_FULL
    _same_bits(g, ref)
    # a second change on top of the first (the list now has 64-bit keys: the appended nodes make the elements wide), then a third
    for where in (0.7, 0.2):
        v3, t3, d2 = synthetic_cut(g.verts, g.tets, axis=(axis + 1) % 3, where=where, every_changed=2)
        g.resync_delta(d2, fixed)
        assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED
        assert np.array_equal(g.tets, t3) and np.array_equal(g.verts, v3)
    ref.resync(v3, t3, fixed)
    _same_bits(g, ref, seed=6)
    g.close()
    ref.close()


def test_delta_resync_partial_changes_and_new_constraints(gpu):
    """removals only, additions only (no new node), changes only, nothing at all; the constrained DOFs follow the call's list"""
    v, t, fixed = _cube(9)
    g = FemIntegrator(v, t, fixed)
    ref = FemIntegrator(v, t, fixed)
    rng = np.random.default_rng(11)
    empty = dict(removed=[], changed_ids=[], changed_nodes=[], added=[], new_xyz=[])
    cur_v, cur_t = v, t
    fixed2 = fixed_vertices_to_dofs(np.concatenate([cube_fixed_plane_i0(9, 9), [400, 401]]))
    for step, fx in (("removed", fixed), ("added", fixed2), ("changed", fixed2), ("none", fixed)):
        d = dict(empty)
        if step == "removed":
            d["removed"] = np.sort(rng.choice(len(cur_t), 40, replace=False)).astype(np.int32)
        elif step == "added":
            d["added"] = cur_t[rng.choice(len(cur_t), 25, replace=False)][:, [1, 0, 2, 3]]   # (duplicates of live elements, mirrored)
        elif step == "changed":
            ids = np.sort(rng.choice(len(cur_t), 30, replace=False)).astype(np.int32)
            d["changed_ids"] = ids
            d["changed_nodes"] = cur_t[ids][:, [0, 2, 1, 3]]
        g.resync_delta(d, fx)
        assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED
        cur_v, cur_t = apply_delta(cur_v, cur_t, {k: np.asarray(x) for k, x in d.items()})
        ref.resync(cur_v, cur_t, fx)
        _same_bits(g, ref, load=-300.0)
        q = g.get_q_state()[0]
        assert not q[fx].any() and np.abs(q).max() > 0
    g.close()
    ref.close()


def test_delta_resync_with_the_node_order_switched_off_and_forced_rebuild(gpu, monkeypatch):
    """FB_RENUMBER_OFF above the size where AUTO looks at the numbering: the list is updated (64-bit keys), bit for bit; with
    FEMBRAIN_RESYNC_DELTA=rebuild the full builder runs from the device copy of the mesh, bit for bit as well"""
    v, t, fixed = _cube(22)
    assert len(v) >= 8192
    v2, t2, d = synthetic_cut(v, t, axis=1, where=0.45)
    g = FemIntegrator(v, t, fixed, renumber=fl.FB_RENUMBER_OFF)
    g.resync_delta(d, fixed)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED and not g.renumbering()[0]
    ref = FemIntegrator(v2, t2, fixed, renumber=fl.FB_RENUMBER_OFF)
    _same_bits(g, ref, load=-200.0)
    monkeypatch.setenv("FEMBRAIN_RESYNC_DELTA", "rebuild")
    g2 = FemIntegrator(v, t, fixed, renumber=fl.FB_RENUMBER_OFF)
    g2.resync_delta(d, fixed)
    assert g2.resync_path() == fl.FB_RESYNC_DELTA_REBUILT
    _same_bits(g2, ref, seed=8, load=-200.0)
    for h in (g, g2, ref):
        h.close()


def test_delta_resync_on_a_handle_with_its_own_node_order(gpu):
    """AUTO at 10,648 nodes: the first cut appends nodes, the elements on them are as wide as the mesh, so the full builder is asked
    (from the device copy of the mesh) and renumbers -- bit for bit what fb_fem_resync gives.  The second cut then finds a renumbered
    handle: the new nodes go into its slab order under the cell size it was made with (a full rebuild would derive a new one), the pair
    list is updated.  Same pattern and values in the caller's ids, the step to the solver's tolerance; the elements stay narrow."""
    v, t, fixed = _cube(22)
    g = FemIntegrator(v, t, fixed)
    assert not g.renumbering()[0]
    v2, t2, d = synthetic_cut(v, t, axis=1, where=0.45)
    g.resync_delta(d, fixed)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_REBUILT
    ref = FemIntegrator(v2, t2, fixed)
    on, sc, si = g.renumbering()
    assert on and ref.renumbering() == (on, sc, si) and si < 2000 < sc
    _same_bits(g, ref, load=-200.0)
    v3, t3, d2 = synthetic_cut(v2, t2, axis=0, where=0.6, every_changed=2, stride=5)
    assert 0 < len(d2["new_xyz"]) < len(v2) // 10
    g.resync_delta(d2, fixed)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED
    ref.resync(v3, t3, fixed)
    on, sc, si = g.renumbering()
    ron, rsc, rsi = ref.renumbering()
    assert on and ron and si <= 1.25 * rsi, ((on, sc, si), (ron, rsc, rsi))   # (as narrow as a fresh slab order makes them)
    bp, bc = g.pattern()
    rp, rc = ref.pattern()
    assert np.array_equal(bp, rp) and np.array_equal(bc, rc)
    u = np.random.default_rng(3).normal(size=g.r) * 0.003
    fa, Ka = g.assemble(u)
    fr, Kr = ref.assemble(u)
    assert np.abs(fa - fr).max() <= 1e-12 * np.abs(fr).max()
    assert np.abs(Ka - Kr).max() <= 2e-7 * np.abs(Kr).max()       # (fp32 stored matrix; the diagonal compensation sums in slot order)
    assert np.abs(g.mass() - ref.mass()).max() <= 1e-12 * np.abs(ref.mass()).max()
    for k in range(2):
        its = []
        for h in (g, ref):
            h.set_uniform_force(1, -200.0)
            its.append(h.do_timestep())
        assert abs(its[0] - its[1]) <= 2
        qa, qr = g.get_q_state()[0], ref.get_q_state()[0]
        assert np.abs(qa - qr).max() <= 2e-5 * np.abs(qr).max() and not qa[fixed].any()
    # once a tenth more nodes have come than the order was built for, a change gets a fresh order: the full builder, bit for bit
    v4, t4, d3 = synthetic_cut(v3, t3, axis=2, where=0.3)
    assert len(v4) * 10 > len(v2) * 11
    g.resync_delta(d3, fixed)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_REBUILT
    ref.resync(v4, t4, fixed)
    _same_bits(g, ref, seed=4, load=-200.0)
    # the next full re-sync starts over
    g.resync(v4, t4, fixed)
    assert g.resync_path() == fl.FB_RESYNC_FULL
    _same_bits(g, ref, seed=9, load=-200.0)
    g.close()
    ref.close()


def test_a_handle_that_expects_cuts_merges_its_first_cut(gpu):
    """fb_fem_params.expect_cuts: the internal node order is chosen at creation -- a grid keeps its plane-by-plane
    order -- so the FIRST cut already merges its nodes into it and updates the plan (FB_RESYNC_DELTA_MERGED) where a default handle sends
    the mesh through the full builder; buffers carry a quarter of slack.  Same pattern, values and steps as a handle made from the cut mesh."""
    v, t, fixed = _cube(22)
    g = FemIntegrator(v, t, fixed, expect_cuts=True)
    assert g.renumbering()[0] and g.resync_path() == fl.FB_RESYNC_FULL
    g.set_uniform_force(1, -200.0)
    assert g.do_timestep() > 0
    cv, ct = v, t
    for k, (axis, where) in enumerate(((1, 0.45), (0, 0.62))):
        cv, ct, d = synthetic_cut(cv, ct, axis=axis, where=where, stride=5 + k)   # (fewer than a tenth more nodes in all: the order is kept)
        g.resync_delta(d, fixed)
        assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED, (k, g.resync_path())
        assert not np.any(g.get_q_state()[0])
    ref = FemIntegrator(cv, ct, fixed)
    bp, bc = g.pattern()
    rp, rc = ref.pattern()
    assert np.array_equal(bp, rp) and np.array_equal(bc, rc)
    u = np.random.default_rng(3).normal(size=g.r) * 0.003
    fa, Ka = g.assemble(u)
    fr, Kr = ref.assemble(u)
    assert np.abs(fa - fr).max() <= 1e-12 * np.abs(fr).max() and np.abs(Ka - Kr).max() <= 2e-7 * np.abs(Kr).max()
    for h in (g, ref):
        h.set_uniform_force(1, -200.0)
    its = [g.do_timestep(), ref.do_timestep()]
    qa, qr = g.get_q_state()[0], ref.get_q_state()[0]
    assert abs(its[0] - its[1]) <= 2 and np.abs(qa - qr).max() <= 2e-5 * np.abs(qr).max() and not qa[fixed].any()
    # reserve_nodes / reserve_elements: room asked for by count
    g2 = FemIntegrator(v, t, fixed, expect_cuts=True, reserve_nodes=2 * len(v), reserve_elements=2 * len(t))
    g2.resync_delta(synthetic_cut(v, t, axis=1, where=0.45, stride=3)[2], fixed)
    assert g2.resync_path() == fl.FB_RESYNC_DELTA_MERGED
    g.close(); ref.close(); g2.close()


def test_fresh_order_rule(gpu, monkeypatch):
    """the default: a renumbered handle merges a change that brings fewer than 2 % more nodes than its order was built for, and sends a larger
    one through the full builder (from the device copy of the mesh) for a fresh order -- bit for bit fb_fem_resync"""
    monkeypatch.delenv("FEMBRAIN_FRESH_ORDER_PERCENT")
    v, t, fixed = _cube(22)
    g = FemIntegrator(v, t, fixed, expect_cuts=True)
    v2, t2, d = synthetic_cut(v, t, axis=1, where=0.45, stride=16)       # 166 new nodes: 1.6 %
    assert 0 < len(d["new_xyz"]) * 100 < 2 * len(v)
    g.resync_delta(d, fixed)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED
    v3, t3, d2 = synthetic_cut(v2, t2, axis=0, where=0.62, stride=16)    # ... and as many again: 3.1 % in all
    g.resync_delta(d2, fixed)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_REBUILT
    ref = FemIntegrator(v3, t3, fixed, renumber=fl.FB_RENUMBER_ON)
    _same_bits(g, ref, load=-200.0)
    g.close(); ref.close()


def test_delta_resync_refuses_bad_input_before_anything_changes(gpu):
    v, t, fixed = _cube(8)
    g = FemIntegrator(v, t, fixed)
    ok = dict(removed=[], changed_ids=[], changed_nodes=[], added=[], new_xyz=[])
    bad = [dict(ok, removed=[5, 5]), dict(ok, removed=[7, 3]), dict(ok, removed=[len(t)]), dict(ok, removed=[-1]),
           dict(ok, removed=[4], changed_ids=[4], changed_nodes=[0, 1, 2, 3]), dict(ok, changed_ids=[3], changed_nodes=[0, 1, 2, len(v)]),
           dict(ok, added=[0, 1, 2, len(v) + 1], new_xyz=[0.0, 0.0, 0.0]), dict(ok, added=[0, 1, -2, 3]),
           dict(ok, removed=np.arange(len(t)))]
    for d in bad:
        with pytest.raises(fl.FbError):
            g.resync_delta(d, fixed)
    for fx in ([3, 3], [5, 4], [-1], [3 * len(v)]):
        with pytest.raises(fl.FbError):
            g.resync_delta(ok, fx)
    g.set_uniform_force(1, -100.0)
    assert g.do_timestep() > 0          # still the old mesh, still usable
    # a flat element gets through the id checks and is refused by the rest-state kernel: the handle is unusable until a full re-sync
    with pytest.raises(fl.FbError):
        g.resync_delta(dict(ok, added=[0, 1, 1, 2]), fixed)
    with pytest.raises(fl.FbError):
        g.do_timestep()
    with pytest.raises(fl.FbError):
        g.resync_delta(ok, fixed)
    g.resync(v, t, fixed)
    g.set_uniform_force(1, -100.0)
    assert g.do_timestep() > 0
    g.close()
    # sharded handles and host-built plans say no
    import os
    os.environ["FEMBRAIN_PLAN_DEVICE"] = "0"
    try:
        gh = FemIntegrator(v, t, fixed)
    finally:
        del os.environ["FEMBRAIN_PLAN_DEVICE"]
    with pytest.raises(fl.FbError):
        gh.resync_delta(ok, fixed)
    gh.close()


def test_delta_resync_random_changes_and_a_hub_node(gpu):
    """a sequence of random changes (removals, in-place changes, additions on old and new nodes) on a small cube, bit for bit against the
    full re-sync after every one; and a fan of 5,000 elements on one node -- the run of its diagonal block alone spans three 2,048-entry
    tiles of the pair list -- losing all but a few of them (a tile of the list that keeps nothing)"""
    v, t, fixed = _cube(8)
    g = FemIntegrator(v, t, fixed)
    ref = FemIntegrator(v, t, fixed)
    rng = np.random.default_rng(2024)
    cv, ct = v, t
    for step in range(8):
        n_t, n_v = len(ct), len(cv)
        ids = rng.permutation(n_t)
        n_rem, n_chg, n_add, n_new = (int(x) for x in rng.integers(0, 40, size=4))
        rem = np.sort(ids[:n_rem]).astype(np.int32)
        chg = np.sort(ids[n_rem:n_rem + n_chg]).astype(np.int32)
        new_xyz = cv[rng.integers(0, n_v, size=n_new)] + rng.normal(size=(n_new, 3)) * 0.03
        pool = n_v + n_new

        def random_tets(k):
            out = np.empty((k, 4), np.int32)
            for i in range(k):
                while True:
                    q = rng.choice(pool, 4, replace=False)
                    p = np.concatenate([cv, new_xyz])[q]
                    if abs(np.dot(np.cross(p[1] - p[0], p[2] - p[0]), p[3] - p[0])) > 1e-7:
                        break
                out[i] = q
            return out
        d = dict(removed=rem, changed_ids=chg, changed_nodes=random_tets(n_chg), added=random_tets(n_add), new_xyz=new_xyz)
        g.resync_delta(d, fixed)
        assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED
        cv, ct = apply_delta(cv, ct, d)
        ref.resync(cv, ct, fixed)
        for name in PLAN:
            assert np.array_equal(_device_plan(g, name), _device_plan(ref, name)), (step, name)
        u = rng.normal(size=g.r) * 0.002
        fa, Ka = g.assemble(u)
        fr, Kr = ref.assemble(u)
        assert np.array_equal(fa, fr) and np.array_equal(Ka, Kr), step
    g.close()
    ref.close()
    # the fan
    k = 5000
    pts = rng.normal(size=(3 * k, 3))
    pts /= np.linalg.norm(pts, axis=1)[:, None]
    fv = np.concatenate([np.zeros((1, 3)), pts * (1.0 + rng.uniform(0, 0.2, size=(3 * k, 1)))])
    ft = np.stack([np.zeros(k, np.int32), 1 + 3 * np.arange(k), 2 + 3 * np.arange(k), 3 + 3 * np.arange(k)], axis=1).astype(np.int32)
    vol = np.einsum("ij,ij->i", np.cross(fv[ft[:, 1]] - fv[ft[:, 0]], fv[ft[:, 2]] - fv[ft[:, 0]]), fv[ft[:, 3]] - fv[ft[:, 0]])
    ft = np.ascontiguousarray(ft[np.abs(vol) > 1e-3])
    ffix = fixed_vertices_to_dofs(np.array([1, 2, 3]))
    g = FemIntegrator(fv, ft, ffix, renumber=fl.FB_RENUMBER_OFF)   # (15,001 nodes: AUTO would look at the numbering)
    d = dict(removed=np.arange(5, len(ft) - 5, dtype=np.int32), changed_ids=[], changed_nodes=[], added=[], new_xyz=[])
    g.resync_delta(d, ffix)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED
    v2, t2 = apply_delta(fv, ft, d)
    ref = FemIntegrator(v2, t2, ffix, renumber=fl.FB_RENUMBER_OFF)
    for name in PLAN:
        assert np.array_equal(_device_plan(g, name), _device_plan(ref, name)), name
    d2 = dict(removed=[], changed_ids=[], changed_nodes=[], added=ft[5:3000], new_xyz=[])   # ... and getting them back, appended
    g.resync_delta(d2, ffix)
    v3, t3 = apply_delta(v2, t2, d2)
    ref.resync(v3, t3, ffix)
    for name in PLAN:
        assert np.array_equal(_device_plan(g, name), _device_plan(ref, name)), name
    g.close()
    ref.close()


@pytest.mark.parametrize("renumber", [fl.FB_RENUMBER_OFF, fl.FB_RENUMBER_ON])
def test_delta_resynced_handle_against_the_cpu_oracle(gpu, renumber):
    """the handle after two changes (the second one updates the pair list; with its own node order too) against the CPU oracle (oracle/fem_oracle.c,
    pinned by the reference build) made from the resulting mesh: block pattern exact, warped assembly and a reference-load step to the
    tolerances of tests/test_fem_gpu.py"""
    from oracle.pyoracle import OrcFem
    v, t, fixed = _cube(9)
    # (fp32 values asked for by name, as at the headline size: FB_MATRIX_AUTO would store a mesh this small as fp64.  The second step's
    # iteration count on this sliver mesh moves by a tenth with tolerance-level differences in the first step's solution, see
    # tools/scratch/dbg_pcg.py: the literal solver and the merged one, both within 1e-6, give 400 and 354)
    g = FemIntegrator(v, t, fixed, renumber=renumber, matrix_precision=fl.FB_MATRIX_F32)
    v2, t2, d = synthetic_cut(v, t, axis=1, where=0.4)
    g.resync_delta(d, fixed)
    v3, t3, d2 = synthetic_cut(v2, t2, axis=2, where=0.55, every_changed=2, stride=6)   # (few new nodes: the order is kept)
    assert 0 < len(d2["new_xyz"]) * 10 < len(v2)
    g.resync_delta(d2, fixed)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED and g.renumbering()[0] == (renumber == fl.FB_RENUMBER_ON)
    o = OrcFem(v3, t3)
    o.integrator(fixed)
    obptr, obcol = o.blocks()
    bptr, bcol = g.pattern()
    assert np.array_equal(bptr, obptr) and np.array_equal(bcol, obcol)
    u = np.random.default_rng(5).normal(size=o.r) * 0.01
    fo, _ = o.assemble(u)
    fg, _ = g.assemble(u)
    assert np.abs(fg - fo).max() <= 1e-9 * np.abs(fo).max()
    f = np.zeros(o.r)
    f[1::3] = -10000.0
    for k in range(2):
        o.set_external_forces(f)
        g.set_external_forces(f)
        io, ig = abs(o.step()), g.do_timestep()
        qo, _ = o.get_state()
        qg = g.get_q_state()[0]
        assert abs(io - ig) <= max(3, 0.02 * io), (k, io, ig)
        assert np.abs(qg - qo).max() <= 2e-4 * np.abs(qo).max() and not qg[fixed].any()


@pytest.mark.parametrize("kw", [dict(matrix_precision=fl.FB_MATRIX_F64), dict(integrator=fl.FB_INTEGRATOR_NEWMARK), dict(exact_tangent=1),
                                dict(pcg_variant=fl.FB_PCG_BLOCK_JACOBI), dict(pcg_variant=fl.FB_PCG_REFERENCE, spmv_kernel=fl.FB_SPMV_ROWS)])
def test_delta_resync_on_handles_of_every_kind(gpu, kw):
    """fp64 matrix, Newmark, exact tangent, block-Jacobi, the literal two-launch solver: what a handle keeps beside the plan (Newmark's
    vectors, the tangent's correction records, the inverse blocks) follows a delta re-sync as it follows a full one -- bit for bit"""
    v, t, fixed = _cube(9)
    g = FemIntegrator(v, t, fixed, **kw)
    g.set_uniform_force(1, -300.0)
    g.do_timestep()
    v2, t2, d = synthetic_cut(v, t, axis=2, where=0.45, every_changed=2)
    g.resync_delta(d, fixed)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED
    ref = FemIntegrator(v2, t2, fixed, **kw)
    _same_bits(g, ref, load=-300.0)
    g.close()
    ref.close()


_SIGMA_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, synthetic_cut, truth_cube
n = 22
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
g = FemIntegrator(v, t, fixed, renumber=fl.FB_RENUMBER_ON)
assert g.renumbering()[0]
cv, ct = v, t
for k, (axis, where, stride) in enumerate(((0, 0.6, 6), (2, 0.35, 6))):
    cv, ct, d = synthetic_cut(cv, ct, axis=axis, where=where, every_changed=2, stride=stride)
    assert 0 < len(d["new_xyz"]) and len(cv) * 10 < len(v) * 11, (len(d["new_xyz"]), len(cv), len(v))
    g.resync_delta(d, fixed)
    assert g.resync_path() == fl.FB_RESYNC_DELTA_MERGED, g.resync_path()
    assert np.array_equal(g.tets, ct) and np.array_equal(g.verts, cv)
    own = g.owned_nodes()
    assert len(own) == len(cv) and np.array_equal(np.sort(own), np.arange(len(cv)))          # old_of_new is a permutation
ref = FemIntegrator(v, t, fixed, renumber=fl.FB_RENUMBER_ON)
ref.resync(cv, ct, fixed)
assert ref.resync_path() == fl.FB_RESYNC_FULL and ref.renumbering()[0]
bp, bc = g.pattern()
rp, rc = ref.pattern()
assert np.array_equal(bp, rp) and np.array_equal(bc, rc)
u = np.random.default_rng(3).normal(size=g.r) * 0.003
fa, Ka = g.assemble(u)
fr, Kr = ref.assemble(u)
assert np.abs(fa - fr).max() <= 1e-12 * np.abs(fr).max()
assert np.abs(Ka - Kr).max() <= 2e-7 * np.abs(Kr).max()
assert np.abs(g.mass() - ref.mass()).max() <= 1e-12 * np.abs(ref.mass()).max()
for h in (g, ref):
    h.set_uniform_force(1, -200.0)
its = [g.do_timestep(), ref.do_timestep()]
qa, qr = g.get_q_state()[0], ref.get_q_state()[0]
assert abs(its[0] - its[1]) <= 2 and np.abs(qa - qr).max() <= 2e-5 * np.abs(qr).max() and not qa[fixed].any(), its
print("sigma delta ok", its, g.renumbering(), ref.renumbering())
"""


def test_delta_resync_on_a_node_order_with_the_second_stage(gpu):
    """the merged delta path on a handle whose internal order has the SECOND stage (rows sorted by element count
    inside windows: k_delta_new_counts, k_delta_sigma_keys, the window-key lookup, key_bits = 10 + window bits).  FEMBRAIN_SIGMA is read
    once per process, so the case runs in a process of its own with FEMBRAIN_SIGMA=1 from the start: renumber = ON, two merged deltas that
    add nodes, then pattern (caller ids) exactly, K / f / mass and a step against a handle re-synced with the whole mesh, and the node
    map is a permutation."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FEMBRAIN_SIGMA="1", FEMBRAIN_TIMING="1", FEMBRAIN_FRESH_ORDER_PERCENT="10")
    out = subprocess.run([sys.executable, "-c", _SIGMA_SCRIPT, root], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "sigma delta ok" in out.stdout
    assert "rows sorted by element count inside windows" in out.stderr      # the second stage really ran
