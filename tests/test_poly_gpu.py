"""Parity of the HIP field / classification / tetrahedralizer / marching-cubes surface path (through the C ABI) against the CPU oracle and
the reference's golden sphere mesh.  Integer outputs (flags, configs, tet connectivity) and fp32 field values of
sqrt-free primitives are compared bit-exactly; primitives that go through sqrt / pow are compared to 1e-6."""
import os

import numpy as np
import pytest

from fembrain_amd.blobtree import make_tree, read_blob, sphere_blob
from fembrain_amd.poly import GpuPoly
from oracle.pyfield import OrcPoly

from meshchecks import surface_mesh_checks

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

OF_RIGHT_OP, OF_LEFT_OP, OF_RANGE, OF_UNARY = 1, 2, 4, 8


def _trees():
    pts = [(0, (0.1 * i - 0.3, 0.05 * i, 0.02 * i * i), (0, 0, 0), (0, 0, 0)) for i in range(6)]
    mixed = [(0, (0, 0, 0), (0, 0, 0), (0, 0, 0)), (1, (-0.5, 0.2, 0), (0.6, 0.3, 0.1), (0, 0, 0)), (5, (0.3, -0.4, 0.2), (0, 0, 0), (0.25, 0, 0)),
             (2, (0, 0, -0.6), (0, 1, 0), (0.2, 0.7, 0)), (3, (0.5, 0.5, 0.5), (0, 0, 1), (0.3, 0, 0)), (4, (-0.5, -0.5, 0.4), (1, 0, 0), (0.3, 0, 0)),
             (7, (0.2, 0.7, -0.3), (1.0, 0.8, 0.64), (1.0 / 0.8 ** 4, -2.0 / 0.64, 1.0))]
    return {
        "sphere": sphere_blob(),
        "blend6_noops": make_tree(pts),
        "range_blend": make_tree(pts, [(4, 0, 5, OF_RANGE, 0, 0)]),
        # op0 = union(op1, op2); op1 = dif(prim0, prim1); op2 = smoothdif(op3, prim2); op3 = range(3..6)
        "nested": make_tree(mixed, [(0, 1, 2, OF_LEFT_OP | OF_RIGHT_OP, 0, 0), (2, 0, 1, 0, 0, 0), (3, 3, 2, OF_LEFT_OP, 0, 0), (4, 3, 6, OF_RANGE, 0, 0)]),
        # two range operators under an intersection: the second range inherits the first one's running field
        "two_ranges": make_tree(pts, [(1, 1, 2, OF_LEFT_OP | OF_RIGHT_OP, 0, 0), (4, 0, 2, OF_RANGE, 0, 0), (4, 3, 5, OF_RANGE, 0, 0)]),
        "ricci": make_tree(pts[:2], [(5, 0, 1, 0, 2.0, 0.5)]),
    }


@pytest.mark.parametrize("name", ["sphere", "blend6_noops", "range_blend", "nested", "two_ranges", "ricci"])
def test_field_array_matches_oracle(gpu, name):
    blob = _trees()[name]
    rng = np.random.default_rng(3)
    pts = np.zeros((5000, 4), np.float32)
    pts[:, :3] = rng.uniform(-1.2, 1.2, size=(5000, 3)).astype(np.float32)
    g = GpuPoly(blob)
    got = g.compute_field_array(pts)
    want = OrcPoly(blob).field_array(pts)
    assert np.array_equal(got[:, :3], pts[:, :3])
    if name in ("nested", "ricci"):  # cylinder/disc/ring use sqrt, Ricci uses pow: last-ulp differences allowed
        assert np.abs(got[:, 3] - want[:, 3]).max() <= 2e-6
    else:
        assert np.array_equal(got[:, 3], want[:, 3])
    if name == "sphere":
        assert g.field((0.25, 0, 0)) == np.float32((1 - 0.0625) ** 3)  # SURVEY.md 8c known answer


def test_sphere_tetmesh_matches_reference_golden(gpu):
    gold = np.load(os.path.join(GOLD, "sphere_tets_c0.1.npz"))
    g = GpuPoly(sphere_blob())
    xyz, tets = g.run_tetrahedralizer(0.1)
    assert g.dims == (12, 12, 12) and len(tets) == 3744 == 6 * len(gold["cell_ijk"])
    # bit-exact against the oracle (welded numbering: exclusive scan in grid order)
    o = OrcPoly(sphere_blob())
    oxyz, otets, oc = o.run_tetrahedralizer(0.1)
    assert np.array_equal(tets, otets) and np.array_equal(xyz, oxyz)
    c = g.counts
    assert (c.n_crossed_edges, c.n_surface_cells, c.n_included_cells, c.n_tet_vertices) == \
        (oc["n_crossed_edges"], oc["n_surface_cells"], oc["n_included_cells"], oc["n_tet_vertices"])
    # against the reference's own file: same cells in the same order, same 6-tet corner pattern, same positions
    cell_lo = np.rint((xyz[tets.reshape(-1, 6, 4)[:, 0, 0]] + 0.5) / 0.1).astype(np.int32)  # LBN of every cell
    assert np.array_equal(cell_lo, gold["cell_ijk"])
    for n, key in ((0, "first_cell_xyz"), (len(gold["cell_ijk"]) - 1, "last_cell_xyz")):
        corners = gold[key]  # 8 un-welded corners of that cell in the file
        for ti in range(6):
            want = corners[gold["pattern"][ti]]
            got = xyz[tets[6 * n + ti]]
            assert np.abs(got - want).max() < 1e-6  # the old kernel forms +1 corners as lower+cellsize (7.45e-9 vs 0)


@pytest.mark.parametrize("name,cellsize", [("nested", 0.09), ("two_ranges", 0.07), ("tumor.blob", 0.15), ("complex.blob", 0.12), ("ventricle.blob", 0.11), ("3slabs.blob", 0.13), ("CylinderWithHoles.blob", 0.17)])
def test_grid_classification_bit_exact(gpu, name, cellsize):
    blob = read_blob(os.path.join(GOLD, "blob", name)) if name.endswith(".blob") else _trees()[name]
    g = GpuPoly(blob)
    dims = g.sweep(cellsize)
    o = OrcPoly(blob)
    assert dims == o.grid_dims(cellsize)
    ogrid = o.sweep(cellsize)
    grid = g.read_grid()
    assert np.array_equal(grid[:, :3], ogrid[:, :3])
    assert np.abs(grid[:, 3] - ogrid[:, 3]).max() <= 2e-6
    # classification is compared on points whose field is not within rounding of the iso value
    oc = o.classify()
    c = g.classify()
    flags, cnt, cfg = g.read_classification()
    safe = np.abs(ogrid[:, 3] - 0.5) > 1e-5
    if safe.all():
        assert np.array_equal(flags, o.edge_flags) and np.array_equal(cnt, o.edge_count) and np.array_equal(cfg, o.config)
        assert c.n_crossed_edges == oc["n_crossed_edges"] and c.n_surface_cells == oc["n_surface_cells"]
        assert c.n_included_cells == oc["n_included_cells"] and c.n_tet_vertices == oc["n_tet_vertices"]
        g.tetrahedralize()
        xyz, tets = g.read_tetmesh()
        oxyz, otets = o.tetrahedralize()
        assert np.array_equal(tets, otets) and np.array_equal(xyz, oxyz)
    else:
        assert (flags != o.edge_flags).sum() <= 8 * (~safe).sum()


def test_sphere_256_properties(gpu):
    """BASELINE config 3 at full size: known counts (6,557,600 included cells -> 39,345,600 tets) and
    size-independent invariants of the emitted mesh."""
    g = GpuPoly(sphere_blob())
    dims = g.sweep_grid((-0.5, -0.5, -0.5), 1.0 / 254.0, (256, 256, 256))
    assert dims == (256, 256, 256)
    c = g.classify()
    assert c.n_points == 256 ** 3 and c.n_cells == 255 ** 3
    assert c.n_included_cells == 6557600
    g.tetrahedralize()
    assert g.counts.n_tets == 39345600
    xyz, tets = g.read_tetmesh()
    assert tets.max() == len(xyz) - 1 and tets.min() == 0
    used = np.zeros(len(xyz), bool)
    used[tets.reshape(-1)] = True
    assert used.all()  # every compacted vertex belongs to a tet
    # vertices come out in grid order (z slowest): lexicographic in (z, y, x)
    key = np.rint((xyz + 0.5) * 254).astype(np.int64)
    lin = key[:, 2] * 65536 + key[:, 1] * 256 + key[:, 0]
    assert (np.diff(lin) > 0).all()
    # every tet of a cell has positive or negative but never zero volume, and the 6 tets fill the cell
    sample = tets[:: 4099][:2000]
    p = xyz[sample].astype(np.float64)
    vol = np.abs(np.einsum("ij,ij->i", p[:, 0] - p[:, 3], np.cross(p[:, 1] - p[:, 3], p[:, 2] - p[:, 3]))) / 6
    assert np.allclose(vol, (1.0 / 254) ** 3 / 6, rtol=1e-3)
    # all samples of the stored grid obey inside <=> |p| <= iso distance 0.4542 up to one cell
    grid = g.read_grid()
    rr = np.sqrt((grid[:, :3].astype(np.float64) ** 2).sum(1))
    inside = grid[:, 3] >= 0.5
    assert inside[rr < 0.45].all() and not inside[rr > 0.46].any()


def test_poly_error_paths_and_order(gpu):
    from fembrain_amd import lib as fl
    g = GpuPoly(sphere_blob())
    with pytest.raises(fl.FbError):
        g.classify()                        # nothing swept yet
    with pytest.raises(fl.FbError):
        g.sweep(0.001)                      # GPUPoly::run refuses cellsize < 0.01
    g.sweep(0.2)
    with pytest.raises(fl.FbError):
        g.tetrahedralize()                  # classify first
    c = g.classify()
    assert c.n_points == int(np.prod(g.dims)) and c.n_cells == int(np.prod(np.array(g.dims) - 1))
    g.tetrahedralize()
    assert g.compute_field_array(np.zeros((0, 4), np.float32)).shape == (0, 4)
    bad = sphere_blob()
    bad.prims[0, 1] = 7                     # matrix index out of range
    with pytest.raises(fl.FbError):
        GpuPoly(bad)
    loop = make_tree([(0, (0, 0, 0), (0, 0, 0), (0, 0, 0))] * 2, [(0, 0, 0, OF_LEFT_OP, 0, 0)])  # operator is its own child
    with pytest.raises(fl.FbError):
        GpuPoly(loop)


def test_far_empty_grid_and_full_grid(gpu):
    """Grids with no surface at all: everything outside (no tets) and everything inside (all cells, config 255)."""
    g = GpuPoly(sphere_blob())
    g.sweep_grid((5.0, 5.0, 5.0), 0.1, (9, 8, 7))
    c = g.classify()
    assert (c.n_crossed_edges, c.n_surface_cells, c.n_included_cells, c.n_tet_vertices) == (0, 0, 0, 0)
    g.tetrahedralize()
    xyz, tets = g.read_tetmesh()
    assert len(xyz) == 0 and len(tets) == 0
    g.sweep_grid((-0.05, -0.05, -0.05), 0.02, (6, 5, 7))
    c = g.classify()
    assert c.n_crossed_edges == 0 and c.n_surface_cells == 0
    assert c.n_included_cells == 5 * 4 * 6 and c.n_tet_vertices == 6 * 5 * 7
    flags, cnt, cfg = g.read_classification()
    assert not flags.any() and (cfg == 255).all()
    o = OrcPoly(sphere_blob())
    o.sweep_grid((-0.05, -0.05, -0.05), 0.02, (6, 5, 7))
    o.classify()
    g.tetrahedralize()
    xyz, tets = g.read_tetmesh()
    oxyz, otets = o.tetrahedralize()
    assert np.array_equal(tets, otets) and np.array_equal(xyz, oxyz)


def test_non_multiple_of_64_row_length_grid(gpu):
    """gx not a multiple of 64 and a total that is not a multiple of 64: words straddle rows and planes."""
    blob = _trees()["nested"]
    lo = np.array([-1.1, -0.9, -1.0], np.float32)
    g = GpuPoly(blob)
    g.sweep_grid(lo, 0.061, (37, 29, 33))
    o = OrcPoly(blob)
    og = o.sweep_grid(lo, 0.061, (37, 29, 33))
    grid = g.read_grid()
    safe = np.abs(og[:, 3] - 0.5) > 1e-5
    assert np.abs(grid[:, 3] - og[:, 3]).max() <= 2e-6
    oc = o.classify()
    c = g.classify()
    if safe.all():
        flags, cnt, cfg = g.read_classification()
        assert np.array_equal(flags, o.edge_flags) and np.array_equal(cfg, o.config)
        assert (c.n_crossed_edges, c.n_surface_cells, c.n_included_cells, c.n_tet_vertices) == \
            (oc["n_crossed_edges"], oc["n_surface_cells"], oc["n_included_cells"], oc["n_tet_vertices"])
        g.tetrahedralize()
        xyz, tets = g.read_tetmesh()
        oxyz, otets = o.tetrahedralize()
        assert np.array_equal(tets, otets) and np.array_equal(xyz, oxyz)


@pytest.mark.parametrize("dims,cell", [((64, 19, 23), 0.035), ((128, 11, 9), 0.018), ((192, 7, 5), 0.012)])
def test_row_aligned_grids_classify_without_the_position_masks(gpu, monkeypatch, dims, cell):
    """gx a multiple of 64: k_classify<ROWS64> derives the seven position masks of a word from its row and plane instead of loading them.
    Against the CPU oracle, and bit for bit against the kernel that loads the masks (FEMBRAIN_CLASSIFY_MASKS=1)."""
    blob = _trees()["nested"]
    lo = np.array([-1.12, -0.35, -0.4], np.float32)
    o = OrcPoly(blob)
    og = o.sweep_grid(lo, cell, dims)
    oc = o.classify()
    out = []
    for masks in ("0", "1"):
        monkeypatch.setenv("FEMBRAIN_CLASSIFY_MASKS", masks)
        g = GpuPoly(blob)
        g.sweep_grid(lo, cell, dims)
        c = g.classify()
        flags, cnt, cfg = g.read_classification()
        g.tetrahedralize()
        xyz, tets = g.read_tetmesh()
        out.append((flags, cnt, cfg, xyz, tets, (c.n_crossed_edges, c.n_surface_cells, c.n_included_cells, c.n_tet_vertices)))
    monkeypatch.delenv("FEMBRAIN_CLASSIFY_MASKS")
    for a, b in zip(out[0][:5], out[1][:5]):
        assert np.array_equal(a, b)
    assert out[0][5] == out[1][5] and out[0][5][2] > 0
    if (np.abs(og[:, 3] - 0.5) > 1e-5).all():
        flags, cnt, cfg, xyz, tets, counts = out[0]
        assert np.array_equal(flags, o.edge_flags) and np.array_equal(cfg, o.config)
        assert counts == (oc["n_crossed_edges"], oc["n_surface_cells"], oc["n_included_cells"], oc["n_tet_vertices"])
        oxyz, otets = o.tetrahedralize()
        assert np.array_equal(tets, otets) and np.array_equal(xyz, oxyz)


# ---- marching-cubes surface (GPUPoly::run) ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,cellsize", [("sphere", 0.1), ("blend6_noops", 0.07), ("range_blend", 0.11), ("two_ranges", 0.09), ("nested", 0.08),
                                           ("peanut.blob", 0.15), ("tumor.blob", 0.2)])
def test_surface_matches_oracle(gpu, name, cellsize):
    """Triangle indices bit-exact; vertex positions bit-exact (same fp32 operation order); normals to 1e-6 for sqrt-free
    trees.  Trees with sqrt/pow primitives differ from the oracle by an ulp in the field, which the forward difference
    over delta = 1e-4 amplifies 1e4-fold in the gradient, so their normals are held to 2e-2 only."""
    blob = read_blob(os.path.join(GOLD, "blob", name)) if name.endswith(".blob") else _trees()[name]
    g, o = GpuPoly(blob), OrcPoly(blob)
    g.sweep(cellsize)
    og = o.sweep(cellsize)
    grid = g.read_grid()
    c = g.classify()
    oc = o.classify()
    exact_field = np.array_equal(grid, og)
    if not exact_field and (np.abs(og[:, 3] - 0.5) <= 1e-5).any():
        pytest.skip("a grid sample within rounding of the iso value: the two classifications may differ")
    assert c.n_crossed_edges == oc["n_crossed_edges"]
    c = g.surface()
    xyz, nrm, tri = g.read_surface()
    oxyz, onrm, otri = o.surface()
    assert c.n_surface_vertices == len(oxyz) == c.n_crossed_edges and c.n_surface_indices == 3 * len(otri)
    assert np.array_equal(tri, otri)
    if exact_field:
        assert np.array_equal(xyz, oxyz)
        assert np.abs(nrm - onrm).max() <= (1e-6 if name in ("sphere", "blend6_noops", "range_blend", "two_ranges") else 2e-2)
    else:
        assert np.abs(xyz - oxyz).max() <= 1e-5
        assert np.abs(nrm - onrm).max() <= 2e-2
    lo = np.asarray(blob.bbox[0], np.float64)
    surface_mesh_checks(xyz, nrm, tri, box=None if name == "sphere" or name.endswith(".blob") else (lo, lo + cellsize * (np.array(g.dims) - 1)),
                        smooth=name != "nested")


def test_surface_sphere_256(gpu):
    """BASELINE config 3 grid: the surface is a closed, consistently oriented 2-manifold of genus 0 on the iso-sphere."""
    g = GpuPoly(sphere_blob())
    g.sweep_grid((-0.5, -0.5, -0.5), 1.0 / 254.0, (256, 256, 256))
    c = g.classify()
    c = g.surface()
    assert c.n_surface_vertices == c.n_crossed_edges and c.n_surface_indices % 3 == 0
    xyz, nrm, tri = g.read_surface()
    surface_mesh_checks(xyz, nrm, tri, radius=np.sqrt(1 - 0.5 ** (1 / 3)), tol=2e-4, ntol=5e-2)
    assert len(xyz) - 3 * len(tri) // 2 + len(tri) == 2
    f = g.compute_field_array(np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1))[:, 3]
    assert np.abs(f - 0.5).max() < 1e-3


def test_surface_on_ragged_and_empty_grids(gpu):
    blob = _trees()["two_ranges"]
    lo = np.array([-1.1, -0.9, -1.0], np.float32)
    g, o = GpuPoly(blob), OrcPoly(blob)
    g.sweep_grid(lo, 0.061, (37, 29, 33))
    og = o.sweep_grid(lo, 0.061, (37, 29, 33))
    assert np.array_equal(g.read_grid(), og)
    g.classify(); o.classify()
    g.surface()
    xyz, nrm, tri = g.read_surface()
    oxyz, onrm, otri = o.surface()
    assert np.array_equal(tri, otri) and np.array_equal(xyz, oxyz) and np.abs(nrm - onrm).max() <= 1e-6
    # no surface at all
    g.sweep_grid((5.0, 5.0, 5.0), 0.1, (9, 8, 7))
    g.classify()
    c = g.surface()
    assert (c.n_surface_vertices, c.n_surface_indices) == (0, 0)
    xyz, nrm, tri = g.read_surface()
    assert xyz.shape == (0, 3) and tri.shape == (0, 3)


def test_apply_fem_displacements(gpu):
    from fembrain_amd import lib as fl
    from fembrain_amd.poly import MESH_SURFACE, MESH_TET
    g = GpuPoly(sphere_blob())
    g.sweep(0.1)
    g.classify()
    with pytest.raises(fl.FbError):
        g.apply_fem_displacements(np.zeros(3), MESH_SURFACE)   # m_isValidVertex is false before run()
    g.surface()
    g.tetrahedralize()
    sxyz, _, _ = g.read_surface()
    txyz, _ = g.read_tetmesh()
    rng = np.random.default_rng(5)
    for mesh, rest in ((MESH_SURFACE, sxyz), (MESH_TET, txyz)):
        u = rng.normal(scale=0.01, size=rest.size)
        out = g.apply_fem_displacements(u, mesh)
        assert np.array_equal(out, rest + u.reshape(-1, 3).astype(np.float32))
        with pytest.raises(fl.FbError):
            g.apply_fem_displacements(u[:-3], mesh)
    # rest positions are kept
    assert np.array_equal(g.read_surface()[0], sxyz) and np.array_equal(g.read_tetmesh()[0], txyz)


# ---- instanced nodes -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,cellsize", [("peanutInstanced", 0.12), ("pizaL2P4", 0.45), ("piza4x4", 1.1)])
def test_instanced_models_match_oracle(gpu, name, cellsize):
    """Reference models built from INSTANCE nodes (copies of an operator subtree under another affine map): expanded at
    tree-compile time into ENTER .. LEAVE blocks on the device, recursion in the oracle.  Point primitives only ->
    field values, classification, tet mesh and surface indices bit-exact."""
    blob = read_blob(os.path.join(GOLD, "blob", name + ".blob"))
    g, o = GpuPoly(blob), OrcPoly(blob)
    rng = np.random.default_rng(5)
    lo, hi = blob.bbox
    pts = np.zeros((20000, 4), np.float32)
    pts[:, :3] = rng.uniform(lo - 0.3, hi + 0.3, size=(20000, 3)).astype(np.float32)
    got, want = g.compute_field_array(pts)[:, 3], o.field_array(pts)[:, 3]
    assert (want > 0.5).sum() > 20
    exact = np.array_equal(got, want)
    assert exact or np.abs(got - want).max() <= 2e-6
    g.sweep(cellsize)
    og = o.sweep(cellsize)
    grid = g.read_grid()
    assert np.array_equal(grid, og) if exact else np.abs(grid[:, 3] - og[:, 3]).max() <= 2e-6
    if not np.array_equal(grid, og):
        return
    c, oc = g.classify(), o.classify()
    flags, cnt, cfg = g.read_classification()
    assert np.array_equal(flags, o.edge_flags) and np.array_equal(cfg, o.config)
    g.tetrahedralize()
    xyz, tets = g.read_tetmesh()
    oxyz, otets = o.tetrahedralize()
    assert np.array_equal(tets, otets) and np.array_equal(xyz, oxyz)
    g.surface()
    sx, sn, st = g.read_surface()
    ox, on, ot = o.surface()
    assert np.array_equal(st, ot) and np.array_equal(sx, ox) and np.abs(sn - on).max() <= 2e-2


def test_surface_follows_the_tet_mesh_displacements(gpu):
    """fb_poly_interpolate_displacements: surface vertex = rest + da + t (db - da) with (a, b, t) of its grid edge.
    Bit-exact against the oracle; exact for a rigid translation; an affine displacement field is reproduced."""
    from fembrain_amd import lib as fl
    blob = read_blob(os.path.join(GOLD, "blob", "peanut.blob"))
    g, o = GpuPoly(blob), OrcPoly(blob)
    g.sweep(0.1); o.sweep(0.1)
    g.classify(); o.classify()
    g.surface()
    g.tetrahedralize()
    sx, _, _ = g.read_surface()
    tx, _ = g.read_tetmesh()
    pairs, w = g.read_surface_binding()
    opairs, ow = o.surface_binding()
    assert np.array_equal(pairs, opairs) and np.array_equal(w, ow)
    # the surface vertex really is the weighted point of its edge
    assert np.abs(tx[pairs[:, 0]] + w[:, None] * (tx[pairs[:, 1]] - tx[pairs[:, 0]]) - sx).max() < 1e-6
    rng = np.random.default_rng(9)
    u = rng.normal(scale=0.02, size=tx.shape)
    assert np.array_equal(g.interpolate_displacements(u), o.interpolate_displacements(sx, u))
    shift = np.tile(np.array([0.25, -0.5, 0.125]), (len(tx), 1))
    assert np.array_equal(g.interpolate_displacements(shift), sx + shift[0].astype(np.float32))
    A = np.array([[0.02, 0.01, 0.0], [-0.01, 0.03, 0.005], [0.0, 0.004, -0.02]])
    aff = tx.astype(np.float64) @ A.T + 0.1
    assert np.abs(g.interpolate_displacements(aff) - (sx + (sx.astype(np.float64) @ A.T + 0.1))).max() < 2e-6
    with pytest.raises(fl.FbError):
        g.interpolate_displacements(u[:-1])


@pytest.mark.parametrize("name", ["ventricle", "complex", "CylinderWithHoles", "3slabs", "tumor", "pizaL2P4", "piza4x4"])
def test_sweep_culling_changes_no_value(gpu, name):
    """The sweep skips, per 64-point run, the primitives whose support box misses the run (poly.hip: SegBox, support_boxes);
    computeFieldArray evaluates every primitive at every point.  Same device arithmetic otherwise, so the two must agree
    BIT FOR BIT on every grid sample -- cylinders, discs, cubes, scaled / rotated / translated primitives and instanced
    subtrees included."""
    blob = read_blob(os.path.join(GOLD, "blob", name + ".blob"))
    g = GpuPoly(blob)
    lo, hi = blob.bbox
    cell = float((hi - lo).max()) / 60.0
    g.sweep(cell)
    grid = g.read_grid()
    again = g.compute_field_array(grid)
    assert np.array_equal(again[:, :3], grid[:, :3])
    assert np.array_equal(again[:, 3], grid[:, 3])
    assert (grid[:, 3] > 0).any() and (grid[:, 3] == 0).any()
    # and on a grid that reaches far outside the model, where nearly everything is culled
    pad = np.float32(2.5)
    dims = tuple(int(np.ceil((h - l + 2 * pad) / (2 * cell))) + 1 for l, h in zip(lo, hi))
    g.sweep_grid(lo - pad, 2 * cell, dims)
    grid = g.read_grid()
    assert np.array_equal(g.compute_field_array(grid)[:, 3], grid[:, 3])


def test_off_surface_points_and_fields(gpu):
    """GPUPoly::computeOffSurfacePointsAndFields: v +- len n with the field there; against numpy fp32 + the oracle's field."""
    blob = _trees()["two_ranges"]
    g, o = GpuPoly(blob), OrcPoly(blob)
    g.sweep(0.09)
    g.classify()
    g.surface()
    xyz, nrm, _ = g.read_surface()
    got = g.compute_off_surface_points_and_fields(0.05)
    d = np.float32(0.05) * nrm
    want = np.empty_like(got)
    want[0::2, :3], want[1::2, :3] = xyz + d, xyz - d
    want[:, 3] = o.field_array(np.concatenate([want[:, :3], np.zeros((len(want), 1), np.float32)], 1))[:, 3]
    assert np.array_equal(got, want)
    assert (got[0::2, 3] < 0.5).mean() > 0.95 and (got[1::2, 3] > 0.5).mean() > 0.95   # outside / inside of the iso-surface


def _random_tree(rng, n_prims, with_instances):
    """A random BlobTree in the flat layout: every primitive type with a field, random affine matrix nodes, a random binary /
    unary / range operator structure over them, optionally instances of earlier subtrees and primitives."""
    from fembrain_amd.blobtree import BlobTree, NULL_BLOB
    mtx = [np.eye(4, dtype=np.float64)[:3].reshape(12)]
    P, ops = [], []

    def new_matrix():
        a = rng.normal(size=(3, 3)) * 0.25 + np.eye(3) * rng.uniform(0.7, 1.4)
        t = rng.uniform(-0.6, 0.6, size=3)
        m = np.concatenate([a, t[:, None]], 1).reshape(12)
        mtx.append(m)
        return len(mtx) - 1

    def new_prim():
        t = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 7, 8, 6]))
        p = np.zeros(20)
        p[0] = t
        p[1] = new_matrix() if rng.random() < 0.4 else 0
        p[4:7] = rng.uniform(-0.8, 0.8, size=3)
        d = rng.normal(size=3)
        p[8:11] = d / np.linalg.norm(d)
        if t == 1:
            p[8:11] = p[4:7] + d * 0.6
        p[12:15] = [rng.uniform(0.1, 0.5), rng.uniform(0.2, 0.9), 0.0]
        if t == 7:
            r = rng.uniform(0.5, 1.0)
            p[8:11] = [1.0, r, r * r]
            p[12:15] = [1.0 / r ** 4, -2.0 / r ** 2, 1.0]
        P.append(p)
        return len(P) - 1

    def new_instance(origin, is_op):
        p = np.zeros(20)
        p[0], p[1] = 9, new_matrix()
        p[12:15] = [origin, 0, 1.0 if is_op else 0.0]
        P.append(p)
        return len(P) - 1

    def build(depth):
        """returns (index, is_op)"""
        if depth == 0 or rng.random() < 0.25:
            if with_instances and ops and rng.random() < 0.2:
                done = [i for i, o in enumerate(ops) if o["done"]]
                if done:
                    return new_instance(int(rng.choice(done)), True), False
            if with_instances and P and rng.random() < 0.1:
                plain = [i for i, p in enumerate(P) if p[0] != 9]
                return new_instance(int(rng.choice(plain)), False), False
            return new_prim(), False
        idx = len(ops)
        o = {"type": 0, "lc": 0, "rc": 0, "flags": 0, "res": [0.0, 0.0], "done": False}
        ops.append(o)
        kind = rng.random()
        if kind < 0.2:      # range operator over freshly made consecutive primitives
            first = new_prim()
            last = first
            for _ in range(int(rng.integers(1, 4))):
                last = new_prim()
            o.update(type=int(rng.choice([4, 0])), lc=first, rc=last, flags=4)
        elif kind < 0.3:    # unary warp: passes its child through
            c, cop = build(depth - 1)
            o.update(type=int(rng.choice([9, 10, 11, 12])), lc=c, flags=8 | (2 if cop else 0))
        else:
            o["type"] = int(rng.choice([0, 1, 2, 3, 4, 5]))
            if o["type"] == 5:
                pw = float(rng.choice([2.0, 3.0]))
                o["res"] = [pw, 1.0 / pw]
            l, lop = build(depth - 1)
            r, rop = build(depth - 1)
            o.update(lc=l, rc=r, flags=(2 if lop else 0) | (1 if rop else 0))
        o["done"] = True
        return idx, True

    root, is_op = build(4 if rng.random() < 0.5 else 3)
    while not is_op or root != 0:   # the root must be operator 0
        P.clear(); ops.clear(); del mtx[1:]
        root, is_op = build(4 if rng.random() < 0.5 else 3)
    O = np.zeros((len(ops), 16), np.float32)
    for i, o in enumerate(ops):
        O[i, 0:4] = [o["type"], o["lc"], o["rc"], NULL_BLOB]
        O[i, 4:6], O[i, 7] = o["res"], o["flags"]
        O[i, 8:11], O[i, 12:15] = -50.0, 50.0   # operator boxes: wide open (instance cull never triggers wrongly)
    Pm = np.array(P, np.float32)
    header = np.zeros(12, np.float32)
    header[0:3], header[3], header[4:7], header[7] = -1.5, 1, 1.5, 1
    header[8:12] = [len(P), len(ops), len(mtx), NULL_BLOB]
    return BlobTree(header, O, Pm, np.array(mtx, np.float32))


@pytest.mark.parametrize("seed", range(24))
def test_random_trees_match_oracle(gpu, seed):
    """Randomised BlobTrees (all primitive types, affine matrices, nested binary / unary / range operators, Ricci blends;
    odd seeds add instances of subtrees and of primitives): point fields and a swept grid against the oracle.  The sweep
    runs the culled kernel, computeFieldArray the unculled one -- both must agree with each other bit for bit and with
    the oracle to 4e-6 (sqrt / pow primitives differ by an ulp between the two libm's)."""
    rng = np.random.default_rng(1000 + seed)
    blob = _random_tree(rng, 0, with_instances=bool(seed & 1))
    g, o = GpuPoly(blob), OrcPoly(blob)
    pts = np.zeros((4000, 4), np.float32)
    pts[:, :3] = rng.uniform(-1.6, 1.6, size=(4000, 3)).astype(np.float32)
    got, want = g.compute_field_array(pts)[:, 3], o.field_array(pts)[:, 3]
    assert np.isfinite(want).all()
    assert np.abs(got - want).max() <= 4e-6 * max(1.0, np.abs(want).max())
    g.sweep_grid((-1.5, -1.5, -1.5), 0.1, (31, 30, 29))
    grid = g.read_grid()
    assert np.array_equal(g.compute_field_array(grid)[:, 3], grid[:, 3])
    og = o.sweep_grid((-1.5, -1.5, -1.5), 0.1, (31, 30, 29))
    assert np.abs(grid[:, 3] - og[:, 3]).max() <= 4e-6 * max(1.0, np.abs(og[:, 3]).max())


# ---- vertex colours (FieldComputer::fieldValueAndColor, Polygonizer.cpp:2110-2410) -------------------------------------------
# Parity unpinned by reference outputs: the reference holds no colour fixtures; the oracle restates its two-pass walk.
def _assert_colors(got_f, got_c, want_f, want_c):
    """fields to the usual 4e-6; colours to 2e-5 of their scale, except where a (smooth) difference / union / intersection
    picks the other child because the two candidates are an ulp apart between the two libm's (a handful of points at most)"""
    assert np.isfinite(want_c).all()
    assert np.abs(got_f - want_f).max() <= 4e-6 * max(1.0, np.abs(want_f).max())
    bad = np.abs(got_c - want_c).max(axis=1) > 2e-5 * max(1.0, np.abs(want_c).max())
    assert bad.mean() <= 2e-3, "%d of %d colours differ" % (bad.sum(), len(bad))


@pytest.mark.parametrize("name", ["peanutInstanced", "complex", "pizaL2P4", "piza4x4", "tumor", "3slabs", "CylinderWithHoles"])
def test_field_color_matches_oracle(gpu, name):
    blob = read_blob(os.path.join(GOLD, "blob", name + ".blob"))
    g, o = GpuPoly(blob), OrcPoly(blob)
    lo, hi = blob.bbox
    rng = np.random.default_rng(5)
    pts = np.zeros((6000, 4), np.float32)
    pts[:, :3] = (lo + (hi - lo) * rng.random((6000, 3))).astype(np.float32)
    got, got_c = g.field_color_array(pts)
    want, want_c = o.field_color_array(pts)
    _assert_colors(got[:, 3], got_c, want[:, 3], want_c)
    assert np.array_equal(got[:, 3], g.compute_field_array(pts)[:, 3])   # the colour walk leaves the field alone
    assert np.abs(want_c).max() > 0.0


@pytest.mark.parametrize("seed", range(12))
def test_field_color_random_trees(gpu, seed):
    rng = np.random.default_rng(1000 + seed)
    blob = _random_tree(rng, 0, with_instances=bool(seed & 1))
    blob.prims[:, 16:19] = rng.random((blob.n_prims, 3)).astype(np.float32)
    g, o = GpuPoly(blob), OrcPoly(blob)
    pts = np.zeros((4000, 4), np.float32)
    pts[:, :3] = rng.uniform(-1.6, 1.6, size=(4000, 3)).astype(np.float32)
    got, got_c = g.field_color_array(pts)
    want, want_c = o.field_color_array(pts)
    _assert_colors(got[:, 3], got_c, want[:, 3], want_c)
    assert np.array_equal(got[:, 3], g.compute_field_array(pts)[:, 3])


def test_single_primitive_color(gpu):
    blob = sphere_blob()
    blob.prims[0, 16:19] = [0.25, 0.5, 0.75]
    g = GpuPoly(blob)
    _, c = g.field_color_array([[0.1, 0.2, 0.3, 0.0]])
    assert np.array_equal(c[0], np.float32([0.25, 0.5, 0.75]))


def test_surface_colors(gpu):
    blob = read_blob(os.path.join(GOLD, "blob", "peanutInstanced.blob"))
    g, o = GpuPoly(blob), OrcPoly(blob)
    g.run(0.12)
    v, _, _ = g.read_surface()
    rgba = g.read_surface_colors()
    assert rgba.shape == (len(v), 4) and np.all(rgba[:, 3] == 1.0) and len(v) > 1000
    pts = np.concatenate([v, np.zeros((len(v), 1), np.float32)], axis=1)
    f, want = o.field_color_array(pts)
    _assert_colors(g.compute_field_array(pts)[:, 3], rgba[:, :3], f[:, 3], want)


# ---- multi-GPU field path: z-slabs of one grid -----------------------------------------------------------------------
def _sharded_mesh(blob, lower, cell, dims, world):
    """the ranks of a `world`-GPU run, played one after the other on this GPU: pieces of the tet mesh in rank order"""
    from fembrain_amd.poly import slab_plan
    polys, nvs = [], []
    for rank in range(world):
        g = GpuPoly(blob)
        p0, p1, z_first, z_count, own_planes, own_layers = slab_plan(dims[2], world, rank)
        g.sweep_slab(lower, cell, dims, z_first, z_count)
        g.classify()
        g.tetrahedralize()
        nvs.append(g.slab_counts(p0, own_planes, own_layers)[0])
        polys.append((g, p0, own_planes, own_layers))
    pieces = [g.read_tetmesh_slab(p0, op, ol, sum(nvs[:r])) for r, (g, p0, op, ol) in enumerate(polys)]
    return pieces


@pytest.mark.parametrize("world", [2, 3, 5, 8])
@pytest.mark.parametrize("name", ["sphere", "complex", "peanutInstanced"])
def test_slab_pieces_concatenate_to_the_single_grid_mesh(gpu, name, world):
    """positions, field samples and marks of a slab are computed from GLOBAL indices, so the pieces of all ranks put end to
    end are bit for bit the tet mesh of the one-GPU run (vertex numbers included)"""
    blob = sphere_blob() if name == "sphere" else read_blob(os.path.join(GOLD, "blob", name + ".blob"))
    g = GpuPoly(blob)
    dims = g.sweep(0.043 if name == "sphere" else 0.11)
    lower, _ = blob.bbox
    g.classify()
    g.tetrahedralize()
    xyz, tets = g.read_tetmesh()
    assert dims[2] >= 2 * world and len(tets) > 1000
    pieces = _sharded_mesh(blob, lower, 0.043 if name == "sphere" else 0.11, dims, world)
    assert np.array_equal(np.concatenate([p[0] for p in pieces]), xyz)
    assert np.array_equal(np.concatenate([p[1] for p in pieces]), tets)
    assert sum(len(p[1]) for p in pieces) == g.counts.n_tets and min(len(p[1]) for p in pieces[1:-1] or pieces) >= 0


def test_slab_grid_samples_equal_the_whole_grid(gpu):
    blob = read_blob(os.path.join(GOLD, "blob", "tumor.blob"))
    g = GpuPoly(blob)
    dims = g.sweep(0.09)
    whole = g.read_grid().reshape(dims[2], dims[1], dims[0], 4)
    s = GpuPoly(blob)
    s.sweep_slab(blob.bbox[0], 0.09, dims, 3, dims[2] - 5)
    part = s.read_grid().reshape(dims[2] - 5, dims[1], dims[0], 4)
    assert np.array_equal(part, whole[3:dims[2] - 2])


def test_slab_argument_checks(gpu):
    from fembrain_amd import lib as fl
    blob = sphere_blob()
    g = GpuPoly(blob)
    dims = (12, 12, 12)
    with pytest.raises(fl.FbError):
        g.sweep_slab(blob.bbox[0], 0.1, dims, 5, 8)       # runs past the grid
    g.sweep_slab(blob.bbox[0], 0.1, dims, 4, 6)           # planes 4..9
    g.classify()
    g.tetrahedralize()
    with pytest.raises(fl.FbError):
        g.slab_counts(4, 2, 2)                            # first owned plane needs the plane below it in the slab
    with pytest.raises(fl.FbError):
        g.slab_counts(5, 4, 4)                            # last owned layer 8 needs plane 10
    assert g.slab_counts(5, 3, 3)[1] % 6 == 0


# ---- field semantics (fb_poly_set_field_semantics): the OpenCL kernels' evaluation and the CPU path's primitive box cull ----------
SURFACE_FIXTURES = (("tumor", 1138), ("peanut", 436), ("dumbel", 698), ("dumbelclose", 626), ("eggshell", 816))


@pytest.mark.parametrize("name,n_vertices", SURFACE_FIXTURES)
def test_shipped_reference_surfaces_through_the_c_abi(gpu, name, n_vertices):
    """The reference ships the outputs of its own GPU polygonizer for five models (data/models/blobtree/*.veg; the leading
    vertices are the marching-cubes surface, tests/golden/make_surface_golden.py).  With FB_FIELD_OPENCL the HIP path gives
    every one of those vertices, in the reference's order, to the 6 digits the files print -- and is bit-identical to the
    oracle in the same mode."""
    from fembrain_amd.poly import FIELD_OPENCL
    from oracle.pyfield import FIELD_OPENCL as ORC_OPENCL, field_mode
    gold = np.load(os.path.join(GOLD, "surface_%s.npz" % name))
    blob = read_blob(os.path.join(GOLD, "blob", name + ".blob"))
    cs = float(gold["cellsize"])
    g = GpuPoly(blob)
    g.set_field_semantics(FIELD_OPENCL)
    g.sweep(cs)
    g.classify()
    g.surface()
    pos, nrm, tri = g.read_surface()
    want = gold["vertices"]
    assert len(pos) == len(want) == n_vertices
    assert (np.abs(pos - want) <= 6e-6 * np.maximum(1.0, np.abs(want))).all(), np.abs(pos - want).max()
    r = (blob.bbox[0].astype(np.float64) - gold["lattice_phase"]) / cs   # the grid origin sits on the file's lattice
    assert np.abs(r - np.rint(r)).max() * cs < 2e-5
    with field_mode(ORC_OPENCL):
        o = OrcPoly(blob)
        o.sweep(cs)
        o.classify()
        opos, onrm, otri = o.surface()
    assert np.array_equal(tri, otri) and np.array_equal(pos, opos) and np.abs(nrm - onrm).max() <= 1e-6
    surface_mesh_checks(pos, nrm, tri, smooth=False)
    # triangle connectivity: the faces of the reference's tet mesh that TetGen left as they were handed to it (boundary faces on
    # surface vertices only, make_surface_golden.py) are all in the triangle list -- a third of it, pinned by the reference's file
    have = set(map(tuple, np.sort(tri.reshape(-1, 3).astype(np.int64), axis=1).tolist()))
    kept = gold["kept_triangles"]
    assert len(kept) > 0.25 * len(tri.reshape(-1, 3)) and all(tuple(k) in have for k in kept.tolist())
    g.close()


def _chain_tree(rng, n_levels):
    """random tree the OpenCL traversal route exists for: a chain of binary / unary operators, each with at most one operator
    child (left or right), ending in a binary or range operator over primitives; flags as ModelReader sets them"""
    from fembrain_amd.blobtree import NULL_BLOB, BlobTree
    P, ops = [], []

    def prim():
        p = np.zeros(20)
        p[0] = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 7, 8]))
        p[4:7] = rng.uniform(-0.6, 0.6, 3)
        d = rng.normal(size=3)
        p[8:11] = d / np.linalg.norm(d)
        if p[0] == 1:
            p[8:11] = p[4:7] + p[8:11] * 0.6
        p[12:15] = [rng.uniform(0.1, 0.5), rng.uniform(0.2, 0.9), 0.0]
        if p[0] == 7:
            r = rng.uniform(0.5, 1.0)
            p[8:11] = [1.0, r, r * r]
            p[12:15] = [1.0 / r ** 4, -2.0 / r ** 2, 1.0]
        p[16:20] = [*rng.uniform(0, 1, 3), 1]
        P.append(p)
        return len(P) - 1

    for lvl in range(n_levels):
        o = {"type": int(rng.choice([0, 1, 2, 3, 4, 5])), "lc": 0, "rc": 0, "flags": 0, "res": [0.0, 0.0]}
        if o["type"] == 5:
            o["res"] = [2.0, 0.5]
        ops.append(o)
    for i, o in enumerate(ops):
        last = i == n_levels - 1
        if last:
            if rng.random() < 0.5:
                first = prim()
                lastp = first
                for _ in range(int(rng.integers(1, 4))):
                    lastp = prim()
                o.update(lc=first, rc=lastp, flags=4)
            else:
                o.update(lc=prim(), rc=prim())
        elif rng.random() < 0.2:   # unary warp over the next operator
            o.update(type=int(rng.choice([9, 10, 11, 12])), lc=i + 1, flags=8 | 2)
        elif rng.random() < 0.5:
            o.update(lc=i + 1, rc=prim(), flags=2)
        else:
            o.update(lc=prim(), rc=i + 1, flags=1)
            ops[i + 1]["flags"] |= 16   # ofIsRightOp (ReadSceneModel.cpp:486)
    O = np.zeros((len(ops), 16), np.float32)
    for i, o in enumerate(ops):
        O[i, 0:4] = [o["type"], o["lc"], o["rc"], NULL_BLOB]
        O[i, 4:6], O[i, 7] = o["res"], o["flags"]
        O[i, 8:11], O[i, 12:15] = -50.0, 50.0
    header = np.zeros(12, np.float32)
    header[0:3], header[3], header[4:7], header[7] = -1.5, 1, 1.5, 1
    header[8:12] = [len(P), len(ops), 1, NULL_BLOB]
    return BlobTree(header, O, np.array(P, np.float32), np.eye(4, dtype=np.float32)[:3].reshape(1, 12))


@pytest.mark.parametrize("seed", range(12))
def test_opencl_semantics_match_oracle_on_random_chains(gpu, seed):
    from fembrain_amd.poly import FIELD_CPU, FIELD_OPENCL
    from oracle.pyfield import FIELD_OPENCL as ORC_OPENCL, cl_has_route, field_mode
    rng = np.random.default_rng(7000 + seed)
    blob = _chain_tree(rng, 1 + seed % 5)
    assert cl_has_route(blob)
    g = GpuPoly(blob)
    pts = np.zeros((4000, 4), np.float32)
    pts[:, :3] = rng.uniform(-1.6, 1.6, size=(4000, 3)).astype(np.float32)
    cpu_before = g.compute_field_array(pts)[:, 3]
    g.set_field_semantics(FIELD_OPENCL)
    with field_mode(ORC_OPENCL):
        o = OrcPoly(blob)
        want = o.field_array(pts)[:, 3]
        og = o.sweep_grid((-1.5, -1.5, -1.5), 0.1, (31, 30, 29))
    got = g.compute_field_array(pts)[:, 3]
    assert np.isfinite(want).all()
    assert np.abs(got - want).max() <= 4e-6 * max(1.0, np.abs(want).max())
    g.sweep_grid((-1.5, -1.5, -1.5), 0.1, (31, 30, 29))
    grid = g.read_grid()
    assert np.array_equal(g.compute_field_array(grid)[:, 3], grid[:, 3])  # culled sweep kernel == unculled point kernel
    assert np.abs(grid[:, 3] - og[:, 3]).max() <= 4e-6 * max(1.0, np.abs(og[:, 3]).max())
    # and back: the default semantics are what they were
    g.set_field_semantics(FIELD_CPU)
    assert np.array_equal(g.compute_field_array(pts)[:, 3], cpu_before)
    assert np.abs(cpu_before - OrcPoly(blob).field_array(pts)[:, 3]).max() <= 4e-6 * max(1.0, np.abs(cpu_before).max())
    g.close()


def test_opencl_semantics_refuse_trees_the_reference_route_builder_cannot_walk(gpu):
    from fembrain_amd import lib as fl
    from fembrain_amd.poly import FIELD_OPENCL
    blob = read_blob(os.path.join(GOLD, "blob", "complex.blob"))   # operators with two operator children
    g = GpuPoly(blob)
    pts = np.zeros((100, 4), np.float32)
    pts[:, :3] = np.random.default_rng(5).uniform(-1, 1, (100, 3))
    before = g.compute_field_array(pts)
    with pytest.raises(fl.FbError, match="traversal"):
        g.set_field_semantics(FIELD_OPENCL)
    assert np.array_equal(g.compute_field_array(pts), before)   # the handle kept its semantics
    # colours exist for the default semantics only
    g2 = GpuPoly(read_blob(os.path.join(GOLD, "blob", "tumor.blob")))
    g2.set_field_semantics(FIELD_OPENCL)
    with pytest.raises(fl.FbError, match="colour"):
        g2.field_color_array(pts)
    g.close(); g2.close()


@pytest.mark.parametrize("name", ["tumor", "ventricle", "complex", "peanutInstanced", "pizaL2P4", "peanut"])
def test_cpu_box_cull_semantics_match_oracle(gpu, name):
    """FB_FIELD_CPU_BOX = the CPU path with computePrimitiveField's isOutsidePrim cull, per point, boxes from the reader
    (PrepareAllBoxes).  It changes values (a primitive's box is tighter than its support), which is why it is a mode."""
    from fembrain_amd.poly import FIELD_CPU_BOX
    from oracle.pyfield import FIELD_CPU_BOX as ORC_BOX, field_mode
    blob = read_blob(os.path.join(GOLD, "blob", name + ".blob"))
    lo, hi = blob.bbox
    rng = np.random.default_rng(11)
    pts = np.zeros((6000, 4), np.float32)
    pts[:, :3] = rng.uniform(lo - 0.3, hi + 0.3, size=(6000, 3)).astype(np.float32)
    g = GpuPoly(blob)
    plain = g.compute_field_array(pts)[:, 3]
    g.set_field_semantics(FIELD_CPU_BOX)
    got = g.compute_field_array(pts)[:, 3]
    cs = float((hi - lo).max()) / 40
    dims = g.sweep(cs)
    grid = g.read_grid()
    with field_mode(ORC_BOX, blob):
        o = OrcPoly(blob)
        want = o.field_array(pts)[:, 3]
        og = o.sweep(cs)
    assert dims == tuple(int(x) for x in o.g)
    tol = 4e-6 * max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() <= tol and np.abs(grid[:, 3] - og[:, 3]).max() <= tol
    assert np.array_equal(g.compute_field_array(grid)[:, 3], grid[:, 3])
    if name in ("tumor", "ventricle", "peanut"):
        assert (got != plain).any()   # the cull is not a no-op on blended primitives
    g.close()
