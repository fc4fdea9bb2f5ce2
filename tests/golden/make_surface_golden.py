"""Generates tests/golden/surface_<model>.npz from the reference's own data files
/root/reference/data/models/blobtree/{tumor,peanut,dumbel,dumbelclose,eggshell}.veg.

Those files are TetGen meshes the application wrote after ``GPUPoly::run`` (src/main.cpp:782-830:
``readbackMeshV3T3`` -> ``TetGenExporter::tesselate`` -> .veg): TetGen keeps its input points first, so the leading
vertices of each file are the marching-cubes surface vertices of the reference's OpenCL polygonizer, in its output order
(exclusive scan of the per-point edge counts in grid order, X then Y then Z edge: data/opencl/Polygonizer.cl:1429-1561).
A surface vertex lies on a grid edge: two of its coordinates are on the lattice ``bbox.lo + k * cellsize``, the third is
the linear root.  Nothing of ours is used to pick them: the lattice is inferred from the file alone (coordinate values
that repeat), and the surface vertices are the leading run with >= 2 lattice coordinates.

TetGen also keeps the input TRIANGLES where it did not refine the boundary: a face that belongs to exactly one tet of the file
and whose three corners are all among those leading surface vertices is a triangle of the marching-cubes surface the reference
handed to TetGen (about a third of them survive the quality refinement).  They are stored as sorted vertex-id triples
(``kept_triangles``): a reference-held pin of the triangle connectivity, as a subset.

Stored (data only): the leading surface vertices as printed (%g, 6 significant digits), the inferred cellsize and
lattice phase per axis, the surviving surface triangles, and the file's vertex / element totals.  Run here once; the .npz files are committed, the
reference tree is not needed at test time."""
import os
import sys
from collections import Counter

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data/models/blobtree/"
MODELS = ("tumor", "peanut", "dumbel", "dumbelclose", "eggshell")


def read_veg_vertices(path):
    lines = open(path).read().split("\n")
    i = lines.index("*VERTICES")
    n = int(lines[i + 1].split()[0])
    v = np.array([[float(x) for x in ln.split()[1:4]] for ln in lines[i + 2:i + 2 + n]])
    j = next(k for k, ln in enumerate(lines) if ln.startswith("*ELEMENTS"))
    m = int(lines[j + 2].split()[0])
    t = np.array([[int(x) - 1 for x in ln.split()[1:5]] for ln in lines[j + 3:j + 3 + m]], np.int64)  # 1-indexed in the file
    return v, m, t


def boundary_faces(t):
    """faces that belong to exactly one tet, as sorted vertex-id triples"""
    f = np.sort(np.concatenate([t[:, [0, 1, 2]], t[:, [0, 1, 3]], t[:, [0, 2, 3]], t[:, [1, 2, 3]]]), axis=1)
    u, c = np.unique(f, axis=0, return_counts=True)
    return u[c == 1]


def busy_gap(values):
    """commonest gap between the busiest coordinate values of one column (lattice values repeat -- every grid line carries
    many vertices --, roots do not)"""
    c = Counter(np.round(values, 5).tolist())
    top = max(c.values())
    busy = np.array(sorted(v for v, k in c.items() if k >= 0.5 * top))
    return Counter(np.round(np.diff(busy), 3).tolist()).most_common(1)[0][0]


def lattice_phase(values, step):
    """least-squares phase (and refined step) of the lattice ``phase + k * step`` through the repeating values of a column"""
    c = Counter(np.round(values, 5).tolist())
    top = max(c.values())
    rep = np.array(sorted(v for v, k in c.items() if k >= max(4, 0.2 * top)))
    anchor = max(c, key=c.get)
    for _ in range(3):
        k = np.rint((rep - anchor) / step)
        keep = np.abs(rep - anchor - k * step) < 0.02 * step
        A = np.stack([np.ones(keep.sum()), k[keep]], 1)
        (phase, step), *_ = np.linalg.lstsq(A, rep[keep], rcond=None)
        anchor = phase
    return float(step), float(phase)


def on_lattice(v, step, phase, tol):
    r = (v - phase) / step
    return np.abs(r - np.rint(r)) * step < tol


for name in MODELS:
    v, n_tets, tets = read_veg_vertices(REF + name + ".veg")
    gaps = [busy_gap(v[:, a]) for a in range(3)]
    gap = min(gaps)  # an axis with few busy grid lines may show a multiple of the cell size
    assert all(abs(g / gap - round(g / gap)) < 1e-3 for g in gaps), (name, gaps)
    steps, phases = zip(*(lattice_phase(v[:, a], gap) for a in range(3)))
    cellsize = float(np.median(steps))
    assert max(abs(s - cellsize) for s in steps) < 2e-5, (name, steps)
    hits = np.stack([on_lattice(v[:, a], cellsize, phases[a], 2e-5 + 1e-5 * np.abs(v[:, a]).max()) for a in range(3)], 1).sum(1)
    lead = int(np.argmax(hits < 2)) if (hits < 2).any() else len(v)
    surf = v[:lead]
    # which axis carries the root: the off-lattice one (vertices with all three on the lattice: the root fell on a grid point)
    bnd = boundary_faces(tets)
    kept = bnd[bnd.max(axis=1) < lead].astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "surface_%s.npz" % name), vertices=surf, cellsize=np.float64(round(cellsize, 4)),
                        lattice_phase=np.array(phases), n_file_vertices=np.int64(len(v)), n_file_tets=np.int64(n_tets), kept_triangles=kept)
    print("%-12s cellsize %.4f (fit %.6f)  surface vertices %d of %d  tets %d  boundary faces %d, on surface vertices only %d"
          % (name, round(cellsize, 4), cellsize, lead, len(v), n_tets, len(bnd), len(kept)))
