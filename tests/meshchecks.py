"""Geometry checks shared by the CPU and GPU polygonizer tests."""
import numpy as np


def surface_mesh_checks(xyz, nrm, tri, radius=None, tol=4e-3, ntol=2e-2, box=None, smooth=True):
    """size-independent properties of a marching-cubes surface of a closed body: closed, consistently oriented 2-manifold
    (every directed edge exactly once, its reverse exactly once), unit normals agreeing with the triangle winding."""
    assert tri.max() == len(xyz) - 1 and len(np.unique(tri)) == len(xyz)  # every vertex used
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]).astype(np.int64)
    key = e[:, 0] * (len(xyz) + 1) + e[:, 1]
    rkey = e[:, 1] * (len(xyz) + 1) + e[:, 0]
    assert len(np.unique(key)) == len(key)
    if box is None:
        assert np.array_equal(np.sort(key), np.sort(rkey))
    else:  # a surface that leaves the swept grid is cut there: open edges only on the grid's outer planes
        lo, hi = np.asarray(box[0], np.float64), np.asarray(box[1], np.float64)
        openv = e[~np.isin(key, rkey)].reshape(-1)
        q = xyz[openv].astype(np.float64)
        assert (np.minimum(np.abs(q - lo), np.abs(q - hi)).min(1) < 1e-5).all()
    assert np.abs(np.linalg.norm(nrm.astype(np.float64), axis=1) - 1).max() < 1e-5
    p = xyz.astype(np.float64)
    fn = np.cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]])
    big = np.linalg.norm(fn, axis=1) > 1e-9
    s = np.sign((fn[big] * nrm[tri[big, 0]]).sum(1))
    if smooth:  # min/max operators make creases where the finite-difference normal is not the face normal
        assert (s == s[0]).all()
    if radius is not None:
        r = np.linalg.norm(p, axis=1)
        assert np.abs(r - radius).max() < tol
        assert np.abs((p / r[:, None] - nrm)).max() < ntol
    return int(s[0])
