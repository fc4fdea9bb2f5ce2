"""Synthetic tet meshes of the reference's own shapes (host-side input generators).

``truth_cube`` follows ``VolMeshSamples::CreateTruthCube`` (reference
src/deformable/VolMeshSamples.cpp:67-130): nodes in i-major order ``i*ny*nz + j*nz + k`` starting at
``(-nx/2, 0, -nz/2) * cellsize``, six tets per cell in the fixed corner pattern
(LBN,LTN,RBN,LBF) (RTN,LTN,LBF,RBN) (RTN,LTN,LTF,LBF) (RTN,RBN,LBF,RBF) (RTN,LBF,LTF,RBF) (RTN,LTF,RTF,RBF).
"""
import numpy as np


def truth_cube(nx, ny, nz, cellsize=0.1):
    """Returns (verts float64 [n,3], tets int32 [m,4])."""
    if nx < 2 or ny < 2 or nz < 2:
        raise ValueError("truth cube needs at least 2 nodes per axis")
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    start = np.array([-float(nx) / 2.0, 0.0, -float(nz) / 2.0]) * cellsize
    verts = start[None, :] + np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1).astype(np.float64) * cellsize
    ci, cj, ck = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), np.arange(nz - 1), indexing="ij")
    ci, cj, ck = ci.ravel(), cj.ravel(), ck.ravel()

    def nid(di, dj, dk):
        return (ci + di) * ny * nz + (cj + dj) * nz + (ck + dk)

    LBN, LBF, LTN, LTF = nid(0, 0, 0), nid(0, 0, 1), nid(0, 1, 0), nid(0, 1, 1)
    RBN, RBF, RTN, RTF = nid(1, 0, 0), nid(1, 0, 1), nid(1, 1, 0), nid(1, 1, 1)
    six = [(LBN, LTN, RBN, LBF), (RTN, LTN, LBF, RBN), (RTN, LTN, LTF, LBF),
           (RTN, RBN, LBF, RBF), (RTN, LBF, LTF, RBF), (RTN, LTF, RTF, RBF)]
    tets = np.stack([np.stack(t, axis=1) for t in six], axis=1).reshape(-1, 4)
    return np.ascontiguousarray(verts), np.ascontiguousarray(tets.astype(np.int32))


def cube_fixed_plane_i0(ny, nz):
    """Node ids of the i = 0 plane (the cantilever's clamped face)."""
    return np.arange(ny * nz, dtype=np.int32)


def fixed_vertices_to_dofs(fixed_vertices):
    """``Deformable::FixedVerticesToFixedDOF`` (reference src/deformable/Deformable.cpp:294-314): sort, x3."""
    v = np.sort(np.asarray(fixed_vertices, dtype=np.int32))
    return (3 * v[:, None] + np.arange(3, dtype=np.int32)[None, :]).reshape(-1).astype(np.int32)


def delaunay_jittered(m, jitter=0.35, cellsize=0.1, seed=12345, shuffle=True, max_edge=None):
    """An UNSTRUCTURED synthetic mesh: Delaunay tetrahedra (scipy) of an m^3 lattice whose points are moved by up to `jitter` cells, nodes
    in random order (no grid order to find again); slivers dropped, elements oriented positively.  Hull nodes get 40-60 neighbours where an
    interior node has ~15: what a TetGen mesh of a body looks like to the solver.  max_edge (in cells): also drop the elements with a longer
    edge -- the flat hull elements that join far-apart hull points, which a quality mesher would not produce.
    Returns (vertices, tets, fixed vertices = the slab x < one cell)."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
    pts = (g + rng.uniform(-jitter, jitter, size=g.shape)) * cellsize
    if shuffle:
        pts = pts[rng.permutation(len(pts))]
    tt = Delaunay(pts).simplices.astype(np.int32)
    vol = np.einsum("ij,ij->i", pts[tt[:, 1]] - pts[tt[:, 0]], np.cross(pts[tt[:, 2]] - pts[tt[:, 0]], pts[tt[:, 3]] - pts[tt[:, 0]])) / 6
    keep = np.abs(vol) > 1e-6 * cellsize ** 3
    if max_edge is not None:
        longest = np.zeros(len(tt))
        for i in range(4):
            for j in range(i + 1, 4):
                longest = np.maximum(longest, np.linalg.norm(pts[tt[:, i]] - pts[tt[:, j]], axis=1))
        keep &= longest <= max_edge * cellsize
    tt, vol = tt[keep], vol[keep]
    tt[vol < 0] = tt[vol < 0][:, [0, 2, 1, 3]]
    return pts, np.ascontiguousarray(tt), np.nonzero(pts[:, 0] < cellsize)[0].astype(np.int32)


def read_veg(path):
    """Minimal Vega .veg reader (``*VERTICES`` / ``*ELEMENTS TET``, 1- or 0-indexed as the header row says).

    Format as in reference data/models/beam3/beam3_tet.veg and src/deformable/VolMeshIO.cpp.
    """
    verts, tets = [], []
    mode = None
    with open(path, "r") as f:
        lines = [ln.strip() for ln in f]
    idx = 0
    while idx < len(lines):
        ln = lines[idx]
        idx += 1
        if not ln or ln.startswith("#"):
            continue
        if ln.startswith("*"):
            key = ln.upper()
            if key.startswith("*VERTICES"):
                mode = "v"
                while not lines[idx] or lines[idx].startswith("#"):
                    idx += 1
                idx += 1  # "<n> 3 0 0"
            elif key.startswith("*ELEMENTS"):
                mode = "e"
                while not lines[idx] or lines[idx].startswith("#"):
                    idx += 1
                if lines[idx].upper().startswith("TET"):
                    idx += 1
                while not lines[idx] or lines[idx].startswith("#"):
                    idx += 1
                idx += 1  # "<m> 4 0"
            else:
                mode = None
            continue
        parts = ln.replace(",", " ").split()
        if mode == "v" and len(parts) >= 4:
            verts.append((int(parts[0]), float(parts[1]), float(parts[2]), float(parts[3])))
        elif mode == "e" and len(parts) >= 5:
            tets.append(tuple(int(p) for p in parts[:5]))
    v = np.array(verts, dtype=np.float64)
    e = np.array(tets, dtype=np.int64)
    base = int(v[:, 0].min()) if len(v) else 1
    return np.ascontiguousarray(v[:, 1:4]), np.ascontiguousarray((e[:, 1:5] - base).astype(np.int32))


def apply_delta(v, t, delta):
    """The mesh a topology delta describes (the contract of ``fb_fem_resync_delta``): ``changed`` elements get their new nodes in
    place, ``removed`` elements are erased keeping the order of the rest (``VolMesh::remove_cell_core``: ``m_vCells.erase``, reference
    src/deformable/VolMesh.cpp:630), ``added`` elements and ``new_xyz`` nodes are appended (``insert_node``: ``push_back``, :1083-1088)."""
    t2 = np.array(t, dtype=np.int32, copy=True)
    if len(delta["changed_ids"]):
        t2[np.asarray(delta["changed_ids"], dtype=np.int64)] = np.asarray(delta["changed_nodes"], dtype=np.int32).reshape(-1, 4)
    keep = np.ones(len(t2), dtype=bool)
    keep[np.asarray(delta["removed"], dtype=np.int64)] = False
    added = np.asarray(delta["added"], dtype=np.int32).reshape(-1, 4)
    t2 = np.ascontiguousarray(np.concatenate([t2[keep], added], axis=0).astype(np.int32))
    new_xyz = np.asarray(delta["new_xyz"], dtype=np.float64).reshape(-1, 3)
    v2 = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64), new_xyz], axis=0))
    return v2, t2


def synthetic_cut(v, t, axis=1, where=0.5, every_changed=3, stride=1):
    """A cut-shaped topology change for tests and probes (not the reference's subdivision tables): every element (a, b, c, d) whose
    nodes lie on both sides of the plane ``x[axis] = lo + where * (hi - lo)`` is split in four on a new node m at its centroid:
    (a, b, c, m), (m, b, c, d), (a, m, c, d), (a, b, m, d).  Every ``every_changed``-th of them is changed in place to the first and the
    other three are appended; the others are removed and all four appended (stride: only every stride-th crossing element is cut).
    Returns (v2, t2, delta) with delta = dict(removed,
    changed_ids, changed_nodes, added, new_xyz) -- ids in the old numbering, removed ascending."""
    v = np.asarray(v, dtype=np.float64)
    t = np.asarray(t, dtype=np.int32)
    x = v[:, axis][t]
    c = v[:, axis].min() + where * (v[:, axis].max() - v[:, axis].min())
    hit = np.nonzero((x.min(axis=1) < c) & (x.max(axis=1) > c))[0][::max(1, stride)]
    m = (len(v) + np.arange(len(hit))).astype(np.int32)
    new_xyz = v[t[hit]].mean(axis=1)
    a, b, cc, d = (t[hit, k] for k in range(4))
    chg = (np.arange(len(hit)) % every_changed) == 0 if every_changed else np.zeros(len(hit), dtype=bool)
    four = np.stack([np.stack([a, b, cc, m], axis=1), np.stack([m, b, cc, d], axis=1), np.stack([a, m, cc, d], axis=1),
                     np.stack([a, b, m, d], axis=1)], axis=1)          # [hit, 4, 4]
    take = np.ones((len(hit), 4), dtype=bool)
    take[chg, 0] = False                                                # (the first piece of a changed element stays in its place)
    delta = dict(removed=np.ascontiguousarray(hit[~chg].astype(np.int32)), changed_ids=np.ascontiguousarray(hit[chg].astype(np.int32)),
                 changed_nodes=np.ascontiguousarray(four[chg, 0].astype(np.int32)),
                 added=np.ascontiguousarray(four[take].reshape(-1, 4).astype(np.int32)), new_xyz=np.ascontiguousarray(new_xyz))
    v2, t2 = apply_delta(v, t, delta)
    return v2, t2, delta
