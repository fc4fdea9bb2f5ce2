"""CPU: the field / tetrahedralizer restatement (oracle/field_oracle.c) against the reference's golden sphere mesh
(data/models/sphere/implicit_sphere.veg via tests/golden/sphere_tets_c0.1.npz) and analytic field values."""
import os

import numpy as np

from fembrain_amd.blobtree import make_tree, read_blob, sphere_blob
from oracle.pyfield import OrcPoly

from meshchecks import surface_mesh_checks

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_sphere_golden_cells_pattern_positions():
    g = np.load(os.path.join(GOLD, "sphere_tets_c0.1.npz"))
    o = OrcPoly(sphere_blob())
    assert o.grid_dims(0.1) == (12, 12, 12)
    xyz, tets, counts = o.run_tetrahedralizer(0.1)
    assert counts["n_included_cells"] == 624 and len(tets) == 3744
    inc = np.nonzero(o.inc_cells)[0]
    ijk = np.stack([inc % 11, (inc // 11) % 11, inc // 121], 1)
    assert np.array_equal(ijk, g["cell_ijk"])  # same cells, same (linear) order as the reference file
    # the 6-tet corner pattern: corner k of a cell = 4*dx + 2*dy + dz
    cell0 = tets[:6]
    lo = xyz[cell0[0, 0]]
    corner_of = {}
    for vid in np.unique(cell0):
        d = np.rint((xyz[vid] - lo) / 0.1).astype(int)
        corner_of[int(vid)] = 4 * d[0] + 2 * d[1] + d[2]
    assert [[corner_of[int(x)] for x in row] for row in cell0] == g["pattern"].tolist()
    assert np.abs(xyz[cell0[0]] - g["first_cell_xyz"][g["pattern"][0]]).max() < 1e-6


def test_wyvill_known_answers():
    o = OrcPoly(sphere_blob())
    pts = np.array([[0.25, 0, 0, 0], [0, 0, 0, 0], [1.0, 0, 0, 0], [0.3, 0.4, 0, 0], [2, 2, 2, 0]], np.float32)
    f = o.field_array(pts)[:, 3]
    assert f[0] == np.float32((1 - 0.0625) ** 3) == np.float32(0.823974609)
    assert f[1] == 1.0 and f[2] == 0.0 and f[4] == 0.0
    assert abs(f[3] - (1 - 0.25) ** 3) < 1e-6


def test_operator_semantics():
    pts = [(0, (0.0, 0, 0), (0, 0, 0), (0, 0, 0)), (0, (0.6, 0, 0), (0, 0, 0), (0, 0, 0)), (0, (0, 0.6, 0), (0, 0, 0), (0, 0, 0))]
    q = np.array([[0.3, 0.1, 0, 0]], np.float32)
    single = [OrcPoly(make_tree([p])).field_array(q)[0, 3] for p in pts]
    assert OrcPoly(make_tree(pts[:2], [(0, 0, 1, 0, 0, 0)])).field_array(q)[0, 3] == max(single[0], single[1])
    assert OrcPoly(make_tree(pts[:2], [(1, 0, 1, 0, 0, 0)])).field_array(q)[0, 3] == min(single[0], single[1])
    assert OrcPoly(make_tree(pts[:2], [(4, 0, 1, 0, 0, 0)])).field_array(q)[0, 3] == single[0] + single[1]
    assert OrcPoly(make_tree(pts[:2], [(2, 0, 1, 0, 0, 0)])).field_array(q)[0, 3] == min(single[0], np.float32(1) - single[1])
    # a range operator sums its primitives whatever its type (reference CPU path, Polygonizer.cpp:1968-1983)
    assert OrcPoly(make_tree(pts, [(0, 0, 2, 4, 0, 0)])).field_array(q)[0, 3] == (single[0] + single[1]) + single[2]
    # no operators: blend of everything
    assert OrcPoly(make_tree(pts)).field_array(q)[0, 3] == (single[0] + single[1]) + single[2]


def test_classification_definitions_on_a_blob_file():
    o = OrcPoly(read_blob(os.path.join(GOLD, "blob", "peanut.blob")))
    o.sweep(0.2)
    c = o.classify()
    g = o.g
    f = o.xyzf[:, 3].reshape(g[2], g[1], g[0])
    ins = f >= 0.5
    nx = (ins[:, :, :-1] ^ ins[:, :, 1:]).sum() + (ins[:, :-1, :] ^ ins[:, 1:, :]).sum() + (ins[:-1] ^ ins[1:]).sum()
    assert c["n_crossed_edges"] == nx
    cells = np.zeros((g[2] - 1, g[1] - 1, g[0] - 1), int)
    for k in range(8):
        dx, dy, dz = (k >> 2) & 1, (k >> 1) & 1, k & 1
        cells += ins[dz:g[2] - 1 + dz, dy:g[1] - 1 + dy, dx:g[0] - 1 + dx].astype(int) << k
    assert np.array_equal(cells.reshape(-1), o.config)
    assert c["n_included_cells"] == (cells != 0).sum()
    xyz, tets = o.tetrahedralize()
    assert len(tets) == 6 * c["n_included_cells"] and tets.max() == len(xyz) - 1


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_cube_table_equals_the_reference_table():
    """Both regenerations of the marching-cubes table (oracle C, product host code; Bloomenthal's cube-table walk) hash to
    the reference's own array (tests/golden/mc_table.json, made by make_poly_golden.py from _CellConfigTable.h)."""
    import json
    from fembrain_amd.poly import cube_table as product_table
    from oracle.pyfield import cube_table as oracle_table
    gold = json.load(open(os.path.join(GOLD, "mc_table.json")))
    for tri, nvert in (oracle_table(), product_table()):
        assert _sha(tri) == gold["tri_table_u8_256x16_sha256"]
        assert _sha(nvert) == gold["num_vertices_u8_256_sha256"]
        assert int(nvert.sum()) == gold["total_indices"]


def test_surface_of_the_sphere_is_a_closed_oriented_manifold():
    """parity unpinned by the reference (it holds no marching-cubes output); pinned by geometry instead."""
    o = OrcPoly(sphere_blob())
    o.sweep(0.1)
    c = o.classify()
    xyz, nrm, tri = o.surface()
    assert len(xyz) == c["n_crossed_edges"]
    # Wyvill (1 - r^2)^3 = 0.5
    surface_mesh_checks(xyz, nrm, tri, radius=np.sqrt(1 - 0.5 ** (1 / 3)))
    assert len(xyz) - 3 * len(tri) // 2 + len(tri) == 2  # Euler characteristic of a sphere
    f = o.field_array(np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1))[:, 3]
    assert np.abs(f - 0.5).max() < 0.02


def test_surface_of_blob_files_is_closed():
    for name in ("peanut", "tumor"):
        o = OrcPoly(read_blob(os.path.join(GOLD, "blob", name + ".blob")))
        o.sweep(0.15)
        o.classify()
        xyz, nrm, tri = o.surface()
        surface_mesh_checks(xyz, nrm, tri)


def test_instanced_nodes_reader_and_field():
    """peanutInstanced.blob: union(blend(p0, p1), INSTANCE of that blend moved by t).  The instance is the original
    subtree evaluated at the mapped point (Polygonizer.cpp:1879-1901), so the field is invariant under the shift by t.
    No reference output exists for instanced models: parity unpinned, pinned by this symmetry."""
    b = read_blob(os.path.join(GOLD, "blob", "peanutInstanced.blob"))
    inst = b.prims[2]
    assert inst[0] == 9 and (inst[12], inst[13], inst[14]) == (1.0, 3.0, 1.0) and inst[8] == 4.0  # op #1 (script id 3), BLEND
    t = np.array([-2.522842, -1.193464, -1.396770], np.float32)
    assert np.allclose(b.mtx[int(inst[1])].reshape(3, 4)[:, 3], -t)          # inverse matrix of the translation
    # instance box = the blend's box moved by t; model box = union over primitives
    assert np.allclose(b.prim_boxes[2][0], b.op_boxes[1][0] + t, atol=1e-6) and np.allclose(b.header[0:3], b.prim_boxes[2][0])
    o = OrcPoly(b)
    rng = np.random.default_rng(11)
    lo, hi = b.op_boxes[1]
    p = np.zeros((400, 4), np.float32)
    p[:, :3] = rng.uniform(lo + 0.2, hi - 0.2, size=(400, 3)).astype(np.float32)
    q = p.copy()
    q[:, :3] = p[:, :3] + t
    fp, fq = o.field_array(p)[:, 3], o.field_array(q)[:, 3]
    assert (fp > 0).any()
    assert np.abs(fp - fq).max() < 2e-6          # q - t is p up to fp32 rounding of the shift
    # outside the original's box the instance contributes nothing (isOutsideOp cull), although the Wyvill support
    # (radius 1) reaches beyond the box (offset 0.5): 0.2 outside the low-x face, level with the leftmost point primitive
    c0 = -b.mtx[1].reshape(3, 4)[:, 3] + t     # centre of the instanced copy of primitive 0
    cut = np.array([[lo[0] + t[0] - 0.2, c0[1], c0[2], 0], [lo[0] + t[0] + 0.01, c0[1], c0[2], 0]], np.float32)
    f = o.field_array(cut)[:, 3]
    assert f[0] == 0.0 and f[1] > 0.1


def test_instanced_models_compile_for_the_device():
    import ctypes as C
    from fembrain_amd import lib as fl
    L = fl.lib()
    for name, n_inst in (("peanutInstanced", 1), ("pizaL2P4", 1), ("piza4x4", 16), ("complex", 0)):
        b = read_blob(os.path.join(GOLD, "blob", name + ".blob"))
        assert int((b.prims[:, 0] == 9).sum()) == n_inst
        steps, slots = C.c_int(0), C.c_int(0)
        fl.check(L.fb_poly_compile_info(b.n_ops, fl.fptr(b.ops), b.n_prims, fl.fptr(b.prims), C.byref(steps), C.byref(slots)))
        assert steps.value >= b.n_ops and 1 <= slots.value <= 16
    # an instance of its own ancestor must be refused, not expanded for ever
    b = read_blob(os.path.join(GOLD, "blob", "peanutInstanced.blob"))
    b.prims[2, 12] = 0.0   # the instance now points at the root union, which contains it
    steps, slots = C.c_int(0), C.c_int(0)
    assert L.fb_poly_compile_info(b.n_ops, fl.fptr(b.ops), b.n_prims, fl.fptr(b.prims), C.byref(steps), C.byref(slots)) != 0


def test_color_walk_known_answers():
    """fieldValueAndColor (Polygonizer.cpp:2110-2353) on trees small enough to do by hand: a lone primitive shows its colour;
    a blend of two points 2 f_l c_l + 2 f_r c_r; a union the channel-wise maximum; a difference the child it took."""
    from fembrain_amd.blobtree import make_tree
    a, b = (0, (-0.3, 0, 0), (0, 0, 0), (0, 0, 0)), (0, (0.3, 0, 0), (0, 0, 0), (0, 0, 0))
    pts = np.float32([[0.0, 0.1, 0.0, 0], [-0.3, 0.05, 0.1, 0], [0.4, 0, 0, 0]])
    ca, cb = np.float32([1, 0.25, 0]), np.float32([0, 0.5, 1])
    for optype in (4, 0, 1, 2):   # opBlend, opUnion, opIntersect, opDif
        blob = make_tree([a, b], [(optype, 0, 1, 0, 0, 0)])
        blob.prims[0, 16:19], blob.prims[1, 16:19] = ca, cb
        o = OrcPoly(blob)
        f, c = o.field_color_array(pts)
        assert np.array_equal(f[:, 3], o.field_array(pts)[:, 3])
        one = OrcPoly(make_tree([a]))
        two = OrcPoly(make_tree([b]))
        fa, fb = one.field_array(pts)[:, 3], two.field_array(pts)[:, 3]
        wa, wb = (np.float32(2) * (np.float32(0.5) + fa) - 1)[:, None] * ca, (np.float32(2) * (np.float32(0.5) + fb) - 1)[:, None] * cb
        if optype == 4:
            want = wa + wb
        elif optype == 0:
            want = np.maximum(wa, wb)
        elif optype == 1:
            want = np.minimum(wa, wb)
        else:
            want = np.where((fa == f[:, 3])[:, None], ca, 0) + np.where((1 - fb == f[:, 3])[:, None], cb, 0)
        assert np.array_equal(c, want.astype(np.float32)), optype
    lone = make_tree([a])
    lone.prims[0, 16:19] = ca
    assert np.array_equal(OrcPoly(lone).field_color_array(pts)[1], np.tile(ca, (3, 1)))


# ---- the reference's own shipped polygonizer outputs (tests/golden/surface_*.npz, from data/models/blobtree/*.veg) ----
SURFACE_FIXTURES = ("tumor", "peanut", "dumbel", "dumbelclose", "eggshell")


def _fixture_check(pos, g, bbox_lo):
    """every vertex, in the reference's output order, to the 6 significant digits the .veg files print; the grid origin
    sits on the fixture's lattice"""
    want = g["vertices"]
    assert len(pos) == len(want), (len(pos), len(want))
    tol = 6e-6 * np.maximum(1.0, np.abs(want))
    assert (np.abs(pos - want) <= tol).all(), np.abs(pos - want).max()
    r = (np.asarray(bbox_lo, np.float64) - g["lattice_phase"]) / float(g["cellsize"])
    assert np.abs(r - np.rint(r)).max() * float(g["cellsize"]) < 2e-5


def test_opencl_mode_reproduces_every_shipped_surface_vertex_in_order():
    from oracle.pyfield import FIELD_OPENCL, field_mode
    for name in SURFACE_FIXTURES:
        g = np.load(os.path.join(GOLD, "surface_%s.npz" % name))
        blob = read_blob(os.path.join(GOLD, "blob", name + ".blob"))
        with field_mode(FIELD_OPENCL):
            o = OrcPoly(blob)
            o.sweep(float(g["cellsize"]))
            o.classify()
            pos, nrm, tri = o.surface()
        _fixture_check(pos, g, blob.bbox[0])
        surface_mesh_checks(pos, nrm, tri, smooth=False)


def test_which_reference_path_wrote_the_shipped_files():
    """tumor.blob (one range BLEND) is the same surface in every mode without the box cull; the four two-primitive trees are
    reproduced ONLY by the OpenCL kernel's evaluation (binary operator evaluated as operator type := operator index,
    data/opencl/Polygonizer.cl:825), neither by the CPU semantics nor by the CPU path's primitive box cull."""
    from oracle.pyfield import FIELD_CPU, FIELD_CPU_BOX, FIELD_OPENCL, field_mode
    match = {}
    for name in SURFACE_FIXTURES:
        g = np.load(os.path.join(GOLD, "surface_%s.npz" % name))
        blob = read_blob(os.path.join(GOLD, "blob", name + ".blob"))
        for mode in (FIELD_CPU, FIELD_CPU_BOX, FIELD_OPENCL):
            with field_mode(mode, blob):
                o = OrcPoly(blob)
                o.sweep(float(g["cellsize"]))
                o.classify()
                pos, _, _ = o.surface()
            want = g["vertices"]
            match[name, mode] = len(pos) == len(want) and bool((np.abs(pos - want) <= 6e-6 * np.maximum(1.0, np.abs(want))).all())
    assert all(match[n, FIELD_OPENCL] for n in SURFACE_FIXTURES)
    assert match["tumor", FIELD_CPU] and not match["tumor", FIELD_CPU_BOX]
    for n in ("peanut", "dumbel", "dumbelclose", "eggshell"):
        assert not match[n, FIELD_CPU] and not match[n, FIELD_CPU_BOX]


def test_surviving_triangles_of_the_shipped_tet_meshes_are_in_the_triangle_list():
    """TetGen keeps an input triangle where it does not refine the boundary: every face of the reference's .veg tet mesh that belongs
    to one tet only and has surface vertices only (tests/golden/surface_*.npz: kept_triangles, 29-33 % of the surface) must be a
    triangle of the marching-cubes list -- the part of the triangle connectivity the reference's own files pin."""
    from oracle.pyfield import FIELD_OPENCL, field_mode
    for name in SURFACE_FIXTURES:
        g = np.load(os.path.join(GOLD, "surface_%s.npz" % name))
        blob = read_blob(os.path.join(GOLD, "blob", name + ".blob"))
        with field_mode(FIELD_OPENCL, blob):
            o = OrcPoly(blob)
            o.sweep(float(g["cellsize"]))
            o.classify()
            _, _, tri = o.surface()
        tri = np.sort(tri.reshape(-1, 3).astype(np.int64), axis=1)
        have = set(map(tuple, tri.tolist()))
        kept = g["kept_triangles"]
        assert len(kept) > 0.25 * len(tri)
        assert all(tuple(k) in have for k in kept.tolist()), name


def test_opencl_mode_semantics():
    from oracle.pyfield import FIELD_OPENCL, cl_has_route, field_mode
    pts = [(0, (0.0, 0, 0), (0, 0, 0), (0, 0, 0)), (0, (0.6, 0, 0), (0, 0, 0), (0, 0, 0)), (0, (0, 0.6, 0), (0, 0, 0), (0, 0, 0))]
    q = np.array([[0.3, 0.1, 0, 0]], np.float32)
    single = [OrcPoly(make_tree([p])).field_array(q)[0, 3] for p in pts]
    with field_mode(FIELD_OPENCL):
        # operator 0 is evaluated as type 0 = UNION whatever it says (Polygonizer.cl:825)
        for optype in (0, 1, 2, 3, 4, 5):
            assert OrcPoly(make_tree(pts[:2], [(optype, 0, 1, 0, 0, 0)])).field_array(q)[0, 3] == max(single[0], single[1])
        # range operators are evaluated by their own type (ComputeRangeField): blend sums, union is a max fold, intersection
        # a min fold from 0, everything else 0
        assert OrcPoly(make_tree(pts, [(4, 0, 2, 4, 0, 0)])).field_array(q)[0, 3] == (single[0] + single[1]) + single[2]
        assert OrcPoly(make_tree(pts, [(0, 0, 2, 4, 0, 0)])).field_array(q)[0, 3] == max(single)
        assert OrcPoly(make_tree(pts, [(1, 0, 2, 4, 0, 0)])).field_array(q)[0, 3] == 0.0
        assert OrcPoly(make_tree(pts, [(2, 0, 2, 4, 0, 0)])).field_array(q)[0, 3] == 0.0
        # no operators: primitive 0 only (ComputeField :884)
        assert OrcPoly(make_tree(pts)).field_array(q)[0, 3] == single[0]
    # LinearBlobTree::setTraversalRoute never ends on an operator with two operator children
    assert cl_has_route(read_blob(os.path.join(GOLD, "blob", "CylinderWithHoles.blob")))
    assert not cl_has_route(read_blob(os.path.join(GOLD, "blob", "complex.blob")))
