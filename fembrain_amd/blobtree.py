"""``.blob`` BlobTree reader -> the LinearBlobTree flat arrays the C ABI takes (fb_poly_create).

Follows reference src/implicit/ReadSceneModel.cpp:238-750 (``ModelReader::read / readNode /
readTransformation``: flat INI file, depth-first node numbering, operator flags, primitive parameter packing,
inverse affine matrix per primitive), src/implicit/Polygonizer.cpp:210-540 (``PrepareAllBoxes``: primitive
boxes with offset ISO_VALUE = 0.5, operator boxes, model box) and src/implicit/LinearBlobTree.cpp:43-167
(``LinearBlobTree::load``: header 12 floats, 16 per operator, 20 per primitive, 12 per matrix node).

Instanced nodes (``PrimitiveType=INSTANCE``): res = (array index of the original node, its script id, original is an
operator), dir.x = the original's type code (ReadSceneModel.cpp:649-668, ``setAllInstancedNodes`` :214-236); boxes as
``PrepareAllBoxes`` orders them (Polygonizer.cpp:210-263): primitives, operators, instanced nodes (the original's box
mapped -- lo and hi corner only, as the reference does -- by the instance's forward matrix), operators again.

Not restated: the traversal-route links of ``LinearBlobTree::setTraversalRoute`` (the HIP evaluator compiles its
own evaluation order from lc/rc/flags; the ``next`` field is left at NULL_BLOB).
"""
import re

import numpy as np

NULL_BLOB = float(0xFFFF)
ISO_VALUE = 0.5
FLT_MAX = float(np.finfo(np.float32).max)
FLT_MIN = float(np.finfo(np.float32).tiny)

PRIM_TYPES = {"POINT": 0, "LINE": 1, "CYLINDER": 2, "DISC": 3, "RING": 4, "CUBE": 5, "TRIANGLE": 6,
              "QUADRICPOINT": 7, "NULL": 8, "INSTANCE": 9}
OP_TYPES = {"UNION": 0, "INTERSECTION": 1, "DIFFERENCE": 2, "SMOOTH DIFFERENCE": 3, "BLEND": 4, "RICCI BLEND": 5,
            "FASTQUADRICPOINTSET": 7, "CACHE": 8, "TWIST": 9, "TAPER": 10, "BEND": 11, "SHEAR": 12}
OF_RIGHT_OP, OF_LEFT_OP, OF_RANGE, OF_UNARY, OF_IS_RIGHT, OF_BREAK = 1, 2, 4, 8, 16, 32
UNARY_OPS = {9, 10, 11, 12}

f32 = np.float32


class BlobTree:
    """Flat arrays in the LinearBlobTree layout (all float32)."""

    def __init__(self, header, ops, prims, mtx, prim_boxes=None, op_boxes=None):
        self.header = np.ascontiguousarray(header, dtype=f32).reshape(12)
        self.ops = np.ascontiguousarray(ops, dtype=f32).reshape(-1, 16)
        self.prims = np.ascontiguousarray(prims, dtype=f32).reshape(-1, 20)
        self.mtx = np.ascontiguousarray(mtx, dtype=f32).reshape(-1, 12)
        self.prim_boxes, self.op_boxes = prim_boxes, op_boxes

    @property
    def n_ops(self):
        return len(self.ops)

    @property
    def n_prims(self):
        return len(self.prims)

    @property
    def bbox(self):
        return self.header[0:3].copy(), self.header[4:7].copy()


def _parse_ini(path):
    sections, cur = {}, None
    with open(path, "r") as f:
        for raw in f:
            ln = raw.strip()
            if not ln or ln.startswith(";") or ln.startswith("#"):
                continue
            m = re.match(r"^\[(.*)\]$", ln)
            if m:
                cur = sections.setdefault(m.group(1).strip(), {})
                continue
            if "=" in ln and cur is not None:
                k, v = ln.split("=", 1)
                cur[k.strip()] = v.strip()
    return sections


def _vec(s, n, default=0.0):
    if s is None:
        return [default] * n
    vals = [float(x) for x in re.findall(r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?", s)]
    return (vals + [default] * n)[:n]


def _ints(s):
    return [] if s is None else [int(x) for x in re.findall(r"-?\d+", s)]


def _affine(sec):
    """Forward matrix T * R * S (column-vector convention) of a node, float32 arithmetic."""
    s = _vec(sec.get("AffineScale"), 3, 1.0)
    q = _vec(sec.get("AffineRotate"), 4, 0.0)
    t = _vec(sec.get("AffineTranslate"), 3, 0.0)
    if "AffineRotate" not in sec:
        q = [0.0, 0.0, 0.0, 1.0]
    x, y, z, w = [f32(v) for v in q]
    R = np.eye(4, dtype=f32)
    R[0, 0] = 1 - 2 * (y * y + z * z); R[0, 1] = 2 * (x * y - w * z); R[0, 2] = 2 * (x * z + w * y)
    R[1, 0] = 2 * (x * y + w * z); R[1, 1] = 1 - 2 * (x * x + z * z); R[1, 2] = 2 * (y * z - w * x)
    R[2, 0] = 2 * (x * z - w * y); R[2, 1] = 2 * (y * z + w * x); R[2, 2] = 1 - 2 * (x * x + y * y)
    T = np.eye(4, dtype=f32)
    T[0:3, 3] = np.array(t, dtype=f32)
    S = np.diag(np.array(s + [1.0], dtype=f32)).astype(f32)
    return (T @ R @ S).astype(f32)


def _prim_box(ptype, pos, dirv, res):
    off = f32(ISO_VALUE)
    one = np.ones(3, dtype=f32)
    pos, dirv = np.asarray(pos, f32), np.asarray(dirv, f32)
    lo, hi = np.full(3, FLT_MAX, f32), np.full(3, FLT_MIN, f32)
    if ptype in (0, 8):
        lo, hi = pos - off, pos + off
    elif ptype == 1:
        expand = off * one + f32(3.0) * off * (dirv - pos)
        lo, hi = pos - expand, dirv + expand
    elif ptype in (3, 4):
        radius = f32(res[0]) + off
        expand = (radius + off) * (one - dirv) + off * dirv
        lo, hi = pos - expand, pos + expand
    elif ptype == 2:
        s1 = pos + f32(res[1]) * dirv
        expand = (off + f32(res[0])) * one + f32(0.5) * off * dirv
        lo, hi = pos - expand, s1 + expand
    elif ptype == 5:
        side = f32(res[0]) + off
        lo, hi = pos - side, pos + side
    elif ptype == 6:
        s2 = np.asarray(res, f32)
        lo = np.minimum(np.minimum(pos, dirv), s2) - off
        hi = np.maximum(np.maximum(pos, dirv), s2) + off
    elif ptype == 7:
        wv = f32(dirv[1]) + off
        lo, hi = pos - wv, pos + wv
    elif ptype == 9:
        lo, hi = np.zeros(3, f32), np.zeros(3, f32)
    return lo.astype(f32), hi.astype(f32)


def read_blob(path):
    ini = _parse_ini(path)
    glob = ini.get("Global", {})
    if int(glob.get("FileVersion", "0")) < 1:
        raise ValueError("%s: invalid file version" % path)
    roots = _ints(glob.get("RootIDs"))
    if not roots:
        raise ValueError("%s: no RootIDs" % path)
    prims, ops, mtx, boxm = [], [], [np.eye(4, dtype=f32)[:3].reshape(12)], [np.eye(4, dtype=f32)]
    script2array = {}

    def read_node(nid):
        sec = ini.get("BLOBNODE %d" % nid)
        if sec is None:
            raise ValueError("%s: missing [BLOBNODE %d]" % (path, nid))
        is_op = sec.get("IsOperator", "0").strip() in ("1", "true", "True", "TRUE")
        if is_op:
            idx = len(ops)
            script2array.setdefault(nid, idx)
            op = {"type": OP_TYPES.get(sec.get("OperatorType", ""), 0), "flags": 0, "lc": 0, "rc": 0, "res": [0.0, 0.0, 0.0, 0.0]}
            ops.append(op)
            t = op["type"]
            if t == 5:
                power = float(sec.get("power", 1.0))
                op["res"][0], op["res"][1] = power, 1.0 / power
            elif t in UNARY_OPS:
                op["flags"] |= OF_UNARY
                op["res"][0] = float(sec.get("factor", sec.get("rate", 1.0)))
            use_range = sec.get("ChildrenIDsUseRange", "0").strip() in ("1", "true", "True")
            if use_range:
                rng = _ints(sec.get("ChildrenIDsRange"))
                ids = [read_node(i)[0] for i in range(rng[0], rng[1] + 1)]
                op["lc"], op["rc"] = ids[0], ids[-1]
                op["flags"] |= OF_RANGE
            else:
                ids = _ints(sec.get("ChildrenIDs"))
                binary = not (op["flags"] & OF_UNARY)
                if binary and len(ids) != 2:
                    raise ValueError("%s: binary operator node %d has %d children" % (path, nid, len(ids)))
                lc, lc_op = read_node(ids[0])
                op["lc"] = lc
                if lc_op:
                    op["flags"] |= OF_LEFT_OP
                if binary:
                    rc, rc_op = read_node(ids[1])
                    op["rc"] = rc
                    if rc_op:
                        op["flags"] |= OF_RIGHT_OP
                        ops[rc]["flags"] |= OF_IS_RIGHT
                    if lc_op and rc_op:
                        ops[lc]["flags"] |= OF_BREAK
                        ops[rc]["flags"] |= OF_BREAK
            return idx, True
        idx = len(prims)
        name = sec.get("PrimitiveType", "").strip()
        if name not in PRIM_TYPES:
            raise ValueError("%s: unknown primitive type '%s'" % (path, name))
        pt = PRIM_TYPES[name]
        pos, dirv, res = [0.0] * 3, [0.0] * 3, [0.0] * 3
        if pt == 7:
            pos = _vec(sec.get("position"), 3)
            scale, radius = f32(float(sec.get("scale", 0))), f32(float(sec.get("radius", 0)))
            dirv = [scale, radius, radius * radius]
            res = [scale / (radius * radius * radius * radius), (f32(-2.0) * scale) / (radius * radius), scale]
        elif pt == 0:
            pos = _vec(sec.get("position"), 3)
        elif pt == 1:
            pos, dirv = _vec(sec.get("start"), 3), _vec(sec.get("end"), 3)
        elif pt in (3, 4):
            pos, dirv = _vec(sec.get("position"), 3), _vec(sec.get("direction"), 3)
            res[0] = float(sec.get("radius", 0))
        elif pt == 2:
            pos, dirv = _vec(sec.get("position"), 3), _vec(sec.get("direction"), 3)
            res[0], res[1] = float(sec.get("radius", 0)), float(sec.get("height", 0))
        elif pt == 5:
            pos = _vec(sec.get("position"), 3)
            res[0] = float(sec.get("side", 0))
        elif pt == 6:
            pos, dirv, res = _vec(sec.get("corner0"), 3), _vec(sec.get("corner1"), 3), _vec(sec.get("corner2"), 3)
        elif pt == 9:
            is_op = int(sec.get("OriginalNodeIsOp", 0))
            res = [0.0, float(sec.get("OriginalNodeIndex", 0)), float(is_op)]
            oname = sec.get("OriginalNodeType", "").strip()
            dirv = [float((OP_TYPES if is_op else PRIM_TYPES).get(oname, 0)), 0.0, 0.0]
        color = _vec(sec.get("MtrlDiffused"), 4)
        fwd = _affine(sec)
        im = 0
        if not np.array_equal(fwd, np.eye(4, dtype=f32)):
            im = len(mtx)
            inv = np.linalg.inv(fwd.astype(np.float64)).astype(f32)
            mtx.append(inv[:3].reshape(12))
            boxm.append(fwd)
        prims.append({"type": pt, "im": im, "pos": pos, "dir": dirv, "res": res, "color": color[:3]})
        script2array.setdefault(nid, idx)
        return idx, False

    read_node(roots[0])
    if not prims:
        raise ValueError("%s: no primitives" % path)
    for p in prims:  # setAllInstancedNodes: script id -> array index of the original
        if p["type"] == 9:
            sid = int(p["res"][1])
            if sid not in script2array:
                raise ValueError("%s: instance of unknown node %d" % (path, sid))
            p["res"][0] = float(script2array[sid])
            n_target = len(ops) if int(p["res"][2]) else len(prims)
            if not 0 <= script2array[sid] < n_target:
                raise ValueError("%s: instance of node %d: original index out of range" % (path, sid))
    # primitive boxes (PrepareAllPrimBBoxes) and model box (PrepareAllBoxes tail: hi starts at FLT_MIN, sic)
    pboxes = []
    for p in prims:
        lo, hi = _prim_box(p["type"], p["pos"], p["dir"], p["res"])
        if p["im"] != 0:
            M = boxm[p["im"]]
            a = (M[:3, :3] @ lo + M[:3, 3]).astype(f32)
            b = (M[:3, :3] @ hi + M[:3, 3]).astype(f32)
            lo, hi = np.minimum(a, b), np.maximum(a, b)
        pboxes.append((lo, hi))
    def op_box(i):
        op = ops[i]
        if op["flags"] & OF_RANGE:
            los = [pboxes[k][0] for k in range(op["lc"], op["rc"] + 1)]
            his = [pboxes[k][1] for k in range(op["lc"], op["rc"] + 1)]
            lo, hi = np.minimum.reduce(los), np.maximum.reduce(his)
        else:
            lo, hi = op_box(op["lc"]) if op["flags"] & OF_LEFT_OP else pboxes[op["lc"]]
            if not (op["flags"] & OF_UNARY):
                rlo, rhi = op_box(op["rc"]) if op["flags"] & OF_RIGHT_OP else pboxes[op["rc"]]
                lo, hi = np.minimum(lo, rlo), np.maximum(hi, rhi)
        op["box"] = (lo, hi)
        return lo, hi

    if ops:
        op_box(0)
    if any(p["type"] == 9 for p in prims):  # PrepareAllInstancedNodesBBoxes, then the operator boxes once more
        for i, p in enumerate(prims):
            if p["type"] != 9:
                continue
            origin = int(p["res"][0])
            lo, hi = ops[origin]["box"] if int(p["res"][2]) else pboxes[origin]
            M = boxm[p["im"]]
            a = (M[:3, :3] @ lo + M[:3, 3]).astype(f32)
            b = (M[:3, :3] @ hi + M[:3, 3]).astype(f32)
            pboxes[i] = (np.minimum(a, b), np.maximum(a, b))
        if ops:
            op_box(0)
    mlo, mhi = np.full(3, FLT_MAX, f32), np.full(3, FLT_MIN, f32)
    for lo, hi in pboxes:
        mlo, mhi = np.minimum(mlo, lo), np.maximum(mhi, hi)
    header = np.zeros(12, f32)
    header[0:3], header[3], header[4:7], header[7] = mlo, 1.0, mhi, 1.0
    header[8], header[9], header[10], header[11] = len(prims), len(ops), len(mtx), NULL_BLOB
    P = np.zeros((len(prims), 20), f32)
    for i, p in enumerate(prims):
        P[i, 0], P[i, 1] = p["type"], p["im"]
        P[i, 4:7], P[i, 8:11], P[i, 12:15], P[i, 16:19], P[i, 19] = p["pos"], p["dir"], p["res"], p["color"], 1.0
    O = np.zeros((len(ops), 16), f32)
    for i, op in enumerate(ops):
        O[i, 0], O[i, 1], O[i, 2], O[i, 3] = op["type"], op["lc"], op["rc"], NULL_BLOB
        O[i, 4:7], O[i, 7] = op["res"][:3], op["flags"]
        lo, hi = op.get("box", (np.zeros(3, f32), np.zeros(3, f32)))
        O[i, 8:11], O[i, 11], O[i, 12:15], O[i, 15] = lo, 1.0, hi, 1.0
    return BlobTree(header, O, P, np.stack(mtx), prim_boxes=pboxes, op_boxes=[o.get("box") for o in ops])


def sphere_blob():
    """data/models/blobtree/sphere.blob built in memory: one POINT at the origin, box +-0.5."""
    header = np.array([-0.5, -0.5, -0.5, 1, 0.5, 0.5, 0.5, 1, 1, 0, 1, NULL_BLOB], f32)
    P = np.zeros((1, 20), f32)
    P[0, 16:20] = [0, 0.6, 0, 1]
    return BlobTree(header, np.zeros((0, 16), f32), P, np.eye(4, dtype=f32)[:3].reshape(1, 12))


def make_tree(prims, ops=()):
    """Small in-memory trees for tests: prims = [(type, pos, dir, res)], ops = [(type, lc, rc, flags, res0, res1)]."""
    P = np.zeros((len(prims), 20), f32)
    boxes = []
    for i, (t, pos, dirv, res) in enumerate(prims):
        P[i, 0] = t
        P[i, 4:7], P[i, 8:11], P[i, 12:15], P[i, 19] = pos, dirv, res, 1.0
        boxes.append(_prim_box(t, pos, dirv, res))
    lo = np.minimum.reduce([b[0] for b in boxes])
    hi = np.maximum.reduce([np.maximum(b[1], FLT_MIN) for b in boxes])
    O = np.zeros((len(ops), 16), f32)
    for i, (t, lc, rc, flags, r0, r1) in enumerate(ops):
        O[i, 0:4] = [t, lc, rc, NULL_BLOB]
        O[i, 4], O[i, 5], O[i, 7] = r0, r1, flags
    header = np.zeros(12, f32)
    header[0:3], header[3], header[4:7], header[7] = lo, 1, hi, 1
    header[8:12] = [len(prims), len(ops), 1, NULL_BLOB]
    return BlobTree(header, O, P, np.eye(4, dtype=f32)[:3].reshape(1, 12))
