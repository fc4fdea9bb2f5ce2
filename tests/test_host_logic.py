"""CPU: host-side logic of the product -- the C-ABI library loads and exports every symbol the headers declare,
the per-rank plan (pattern, SELL-64 layout, contribution lists, halo/send lists), mesh generators and readers."""
import ctypes as C
import os
import re
import time

import numpy as np
import pytest

from fembrain_amd import lib as fl
from fembrain_amd.blobtree import read_blob, sphere_blob
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, read_veg, truth_cube
from oracle.pyoracle import OrcFem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    L = C.CDLL(fl.LIB_PATH)
    names = set()
    for hdr in ("fembrain_hip.h", "fembrain_hip_testing.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(fb_[a-z0-9_]+)\s*\(", text))
    assert len(names) >= 50
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    fl.lib()  # binds argument types for all of them


def test_no_device_is_reported_not_faked():
    L = fl.lib()
    if L.fb_device_count() > 0:
        pytest.skip("a GPU is present")
    from fembrain_amd.fem import FemIntegrator
    v, t = truth_cube(3, 3, 3)
    with pytest.raises(fl.FbError) as ei:
        FemIntegrator(v, t, [])
    assert ei.value.code == fl.FB_EDEVICE  # the product path fails loudly, there is no CPU fallback


def test_truth_cube_matches_reference_layout():
    v, t = truth_cube(3, 4, 5, 0.1)
    assert v.shape == (60, 3) and t.shape == (6 * 2 * 3 * 4, 4)
    assert np.allclose(v[0], [-0.15, 0, -0.25]) and np.allclose(v[1] - v[0], [0, 0, 0.1])  # k fastest
    # first cell: LBN=0, LBF=1 (k+1), LTN=nz (j+1), RBN=ny*nz (i+1)
    assert t[0].tolist() == [0, 5, 20, 1] and t[1].tolist() == [25, 5, 1, 20]
    vol = np.abs(np.einsum("ij,ij->i", v[t[:, 0]] - v[t[:, 3]], np.cross(v[t[:, 1]] - v[t[:, 3]], v[t[:, 2]] - v[t[:, 3]]))) / 6
    assert np.allclose(vol.sum(), 0.2 * 0.3 * 0.4)
    assert fixed_vertices_to_dofs([2, 0]).tolist() == [0, 1, 2, 6, 7, 8]


def _plan(v, t, fixed, n_ranks=1, rank=0, splits=None):
    L = fl.lib()
    h = C.c_void_p()
    sp = None if splits is None else np.asarray(splits, np.int32)
    tt = np.ascontiguousarray(t, np.int32).reshape(-1)
    fd = np.ascontiguousarray(fixed, np.int32)
    fl.check(L.fb_plan_create(C.byref(h), len(v), len(t), fl.iptr(tt), len(fd), fl.iptr(fd), n_ranks, rank, fl.iptr(sp)))
    info = np.zeros(12, np.int32)
    L.fb_plan_info(h, fl.iptr(info))

    def get(name):
        cnt = L.fb_plan_get(h, name.encode(), None, 0)
        assert cnt >= 0
        a = np.zeros(cnt, np.int32)
        assert L.fb_plan_get(h, name.encode(), fl.iptr(a), cnt) == cnt
        return a
    keys = ["n_owned", "n_halo", "n_tets", "n_blocks", "n_slices", "n_slots", "n_crows", "n_send", "node_lo", "node_hi", "n_fixed_owned", "n_ranks"]
    return dict(zip(keys, info.tolist())), get, (L, h)


def test_plan_pattern_and_contributions_match_oracle():
    n = 6
    v, t = truth_cube(n, n, n)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    info, get, (L, h) = _plan(v, t, fixed)
    o = OrcFem(v, t)
    obptr, obcol = o.blocks()
    assert np.array_equal(get("bptr"), obptr) and np.array_equal(get("bcol"), obcol)
    assert info["n_fixed_owned"] == len(fixed) and info["n_halo"] == 0 and info["n_tets"] == len(t)
    # SELL layout reproduces the CSR pattern
    so, colidx, blk_slot, bptr, bcol = get("slice_off"), get("colidx"), get("blk_slot"), get("bptr"), get("bcol")
    for a in (0, 1, 63, 64, 100, len(v) - 1):
        for k, p in enumerate(range(bptr[a], bptr[a + 1])):
            assert blk_slot[p] == so[a // 64] + k and colidx[blk_slot[p] * 64 + a % 64] == bcol[p]
    # every (tet, i, j) appears exactly once, in the list of block (tet[i], tet[j]), in ascending element order
    contrib = get("contrib").view(np.uint32)
    coff, ccnt = get("slot_coff"), get("slot_ccnt")
    valid = contrib[contrib != 0xFFFFFFFF]
    assert len(valid) == 16 * len(t) and len(np.unique(valid)) == len(valid)
    for a in (0, 70, 129):
        for k, p in enumerate(range(bptr[a], bptr[a + 1])):
            slot = blk_slot[p]
            lst = [int(contrib[(coff[slot] + r) * 64 + a % 64]) for r in range(ccnt[slot])]
            lst = [c for c in lst if c != 0xFFFFFFFF]
            assert lst == sorted(lst) and lst
            for c in lst:
                e, i, j = c >> 4, (c >> 2) & 3, c & 3
                assert t[e][i] == a and t[e][j] == bcol[p]
    L.fb_plan_destroy(h)


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_sharded_plans_cover_the_mesh_consistently(n_ranks):
    n = 7
    v, t = truth_cube(n, n, n)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    planes = [n * r // n_ranks for r in range(n_ranks + 1)]
    splits = [p * n * n for p in planes]
    plans = [_plan(v, t, fixed, n_ranks, r, splits) for r in range(n_ranks)]
    o = OrcFem(v, t)
    obptr, obcol = o.blocks()
    owned_total, fixed_total = 0, 0
    for r, (info, get, _) in enumerate(plans):
        l2g = get("local2global")
        lo, hi = splits[r], splits[r + 1]
        assert info["node_lo"] == lo and info["node_hi"] == hi and np.array_equal(l2g[:hi - lo], np.arange(lo, hi))
        owned_total += info["n_owned"]
        fixed_total += info["n_fixed_owned"]
        # owned rows carry exactly the global pattern (columns translated back to global ids)
        bptr, bcol = get("bptr"), get("bcol")
        for a in (0, info["n_owned"] // 2, info["n_owned"] - 1):
            assert np.array_equal(l2g[bcol[bptr[a]:bptr[a + 1]]], obcol[obptr[lo + a]:obptr[lo + a + 1]])
        # halo = exactly the non-owned columns, sorted, grouped by owner; send lists mirror the neighbours' halos
        halo = l2g[hi - lo:]
        assert np.array_equal(halo, np.unique(halo)) and not ((halo >= lo) & (halo < hi)).any()
        cols = np.unique(l2g[bcol])
        assert np.array_equal(np.sort(np.concatenate([halo, np.arange(lo, hi)])), np.union1d(cols, np.arange(lo, hi)))
        hoff = get("halo_off")
        for q in range(n_ranks):
            seg = halo[hoff[q]:hoff[q + 1]]
            assert ((seg >= splits[q]) & (seg < splits[q + 1])).all()
            if q != r:
                qinfo, qget, _ = plans[q]
                soff, sloc = qget("send_off"), qget("send_local")
                sent = splits[q] + sloc[soff[r]:soff[r + 1]]  # what q packs for r, as global ids
                assert np.array_equal(sent, seg)
    assert owned_total == len(v) and fixed_total == len(fixed)
    for _, _, (L, h) in plans:
        L.fb_plan_destroy(h)


def test_plan_rejects_bad_input():
    v, t = truth_cube(3, 3, 3)
    L = fl.lib()
    h = C.c_void_p()
    tt = np.ascontiguousarray(t, np.int32).reshape(-1)
    bad = np.array([5, 3], np.int32)
    assert L.fb_plan_create(C.byref(h), len(v), len(t), fl.iptr(tt), 2, fl.iptr(bad), 1, 0, None) == fl.FB_EINVAL
    assert b"ascending" in L.fb_last_error()
    tt2 = tt.copy()
    tt2[7] = 999
    assert L.fb_plan_create(C.byref(h), len(v), len(t), fl.iptr(tt2), 0, None, 1, 0, None) == fl.FB_EINVAL
    assert L.fb_plan_create(C.byref(h), len(v), len(t), fl.iptr(tt), 0, None, 2, 2, None) == fl.FB_EINVAL


def test_blob_reader_and_veg_reader():
    b = read_blob(os.path.join(GOLD, "blob", "sphere.blob"))
    s = sphere_blob()
    assert np.array_equal(b.header, s.header) and b.n_ops == 0 and b.n_prims == 1
    t = read_blob(os.path.join(GOLD, "blob", "tumor.blob"))
    assert (t.n_prims, t.n_ops, len(t.mtx)) == (10, 1, 11)
    assert int(t.ops[0, 0]) == 4 and int(t.ops[0, 7]) & 4  # BLEND over a primitive range
    assert (t.header[0:3] < t.header[4:7]).all()
    c = read_blob(os.path.join(GOLD, "blob", "complex.blob"))
    assert (c.n_prims, c.n_ops) == (12, 11)
    # every operator is referenced once (the reference's CheckForBlobTreeErrors rule)
    refs = []
    for o in c.ops:
        fl_ = int(o[7])
        if fl_ & 4:
            continue
        if fl_ & 2:
            refs.append(int(o[1]))
        if fl_ & 1:
            refs.append(int(o[2]))
    assert sorted(refs) == list(range(1, c.n_ops))
    g = np.load(os.path.join(GOLD, "fem_beam3.npz"))
    assert g["verts"].shape == (208, 3) and g["tets"].max() == 207 and g["tets"].min() == 0


@pytest.mark.parametrize("n_ranks", [1, 2, 3])
def test_contribution_lists_reassemble_the_oracle_matrix(n_ranks):
    """Host emulation of the row-gather assembly (k_assemble_rows) from the plan's contribution lists: for every rank
    and a sample of its owned rows, summing the oracle's element blocks K_e[i][j] over the listed (tet, i, j) gives the
    oracle's assembled global rows -- checks local tet numbering, tet_global and the lists of sharded plans."""
    n = 6
    v, t = truth_cube(n, n, n)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    o = OrcFem(v, t)
    rng = np.random.default_rng(8)
    u = rng.normal(size=o.r) * 0.01
    _, Kglob = o.assemble(u)
    ia, ja = o.csr()
    obptr, obcol = o.blocks()
    planes = [n * r // n_ranks for r in range(n_ranks + 1)]
    splits = [p * n * n for p in planes]
    total_valid = 0
    for r in range(n_ranks):
        info, get, (L, h) = _plan(v, t, fixed, n_ranks, r, splits)
        l2g, tg, lt = get("local2global"), get("tet_global"), get("tets").reshape(-1, 4)
        assert np.array_equal(l2g[lt], t[tg])  # local tets are the global ones, renumbered
        contrib = get("contrib").view(np.uint32)
        total_valid += int((contrib != 0xFFFFFFFF).sum())
        coff, ccnt, bptr, bcol, blk_slot = get("slot_coff"), get("slot_ccnt"), get("bptr"), get("bcol"), get("blk_slot")
        lo = splits[r]
        for a in sorted(set([0, 1, info["n_owned"] // 3, info["n_owned"] - 1])):
            ga = lo + a
            for p in range(bptr[a], bptr[a + 1]):
                slot = blk_slot[p]
                blk = np.zeros((3, 3))
                for rr in range(ccnt[slot]):
                    c = int(contrib[(coff[slot] + rr) * 64 + a % 64])
                    if c == 0xFFFFFFFF:
                        continue
                    e, i, j = c >> 4, (c >> 2) & 3, c & 3
                    _, Ke, _ = o.element(int(tg[e]), u)
                    blk += Ke[3 * i:3 * i + 3, 3 * j:3 * j + 3]
                gb = l2g[bcol[p]]
                pos = int(np.nonzero(obcol[obptr[ga]:obptr[ga + 1]] == gb)[0][0])
                want = np.stack([Kglob[ia[3 * ga + k] + 3 * pos: ia[3 * ga + k] + 3 * pos + 3] for k in range(3)])
                assert np.abs(blk - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
        L.fb_plan_destroy(h)
    assert total_valid == 16 * len(t)  # every (tet, i, j) is assembled by exactly one rank: the owner of node t[i]


def test_haptic_force_spreading_rings():
    """Deformable::applyHapticForces: ring j of the breadth-first walk over mesh edges gets (size - j)/size of the force."""
    from fembrain_amd.fem import spread_haptic_forces
    n = 7
    v, t = truth_cube(n, n, n)
    o = OrcFem(v, t)
    bptr, bcol = o.blocks()
    centre = 3 * n * n + 3 * n + 3
    f = np.zeros(3 * len(v))
    spread_haptic_forces(bptr, bcol, [centre], [(0.0, 5.0, 0.0)], 3, f)
    fy = f[1::3]
    assert fy[centre] == 5.0
    ring1 = set(int(b) for b in bcol[bptr[centre]:bptr[centre + 1]]) - {centre}
    assert len(ring1) == 14 and all(abs(fy[b] - 5.0 * 2 / 3) < 1e-15 for b in ring1)   # 14 edge neighbours in the 6-tet cube
    ring2 = set()
    for b in ring1:
        ring2 |= set(int(c) for c in bcol[bptr[b]:bptr[b + 1]])
    ring2 -= ring1 | {centre}
    assert ring2 and all(abs(fy[b] - 5.0 / 3) < 1e-15 for b in ring2)
    assert np.count_nonzero(fy) == 1 + len(ring1) + len(ring2) and not f[0::3].any() and not f[2::3].any()
    # two haptic vertices superpose
    g = np.zeros(3 * len(v))
    spread_haptic_forces(bptr, bcol, [centre, 0], [(0.0, 5.0, 0.0), (1.0, 0.0, 0.0)], 2, g)
    assert g[3 * centre + 1] == 5.0 and g[0] == 1.0 and abs(g[3 * 1] - 0.5) < 1e-15


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under fembrain_amd/ or include/ may import, include or link it."""
    import subprocess
    bad = []
    for base in ("fembrain_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dirpath or "__pycache__" in dirpath:
                continue
            for fn in files:
                if not fn.endswith((".py", ".h", ".hip", ".cpp", ".c")) and fn != "Makefile":
                    continue
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                for pat in (r"^\s*(from|import)\s+oracle", r"#include\s+[\"<].*oracle", r"libfem_oracle", r"libfem_ref", r"oracle/"):
                    if re.search(pat, text, flags=re.M):
                        bad.append((os.path.join(dirpath, fn), pat))
    assert not bad, bad
    needed = subprocess.check_output(["readelf", "-d", fl.LIB_PATH], text=True)
    assert "oracle" not in needed and "fem_ref" not in needed


def test_plan_is_independent_of_the_builder_threads(monkeypatch):
    """The pattern / SELL / contribution-list phases of the host plan run row-parallel on host threads (a re-sync after a
    cut waits for them); every array must come out identical for any thread count, sharded or not."""
    n = 30   # 27,000 nodes: more than 2048 rows per thread for 7 threads
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    names = ["local2global", "halo_off", "send_off", "send_local", "tets", "tet_global", "bptr", "bcol", "slice_off", "colidx", "blk_slot",
             "slot_coff", "slot_ccnt", "contrib", "dofmask"]
    for n_ranks, rank in ((1, 0), (2, 1)):
        got = []
        for threads in ("1", "7"):
            monkeypatch.setenv("FEMBRAIN_PLAN_THREADS", threads)
            info, get, (L, h) = _plan(v, t, fixed, n_ranks, rank)
            got.append((info, {k: get(k) for k in names}))
            L.fb_plan_destroy(h)
        assert got[0][0] == got[1][0]
        for k in names:
            assert np.array_equal(got[0][1][k], got[1][1][k]), k


def test_cut_create_validates_before_touching_a_device():
    """fb_cut_create checks the tet indices (the reference's createMemBuffers does not) -- host logic, no GPU needed"""
    import ctypes as C
    from fembrain_amd import lib as fl
    L = fl.lib()
    v = np.zeros((4, 3))
    h = C.c_void_p()
    bad = np.array([[0, 1, 2, 4]], np.uint32)
    assert L.fb_cut_create(C.byref(h), 0, 4, fl.dptr(v), 1, fl.uptr(bad)) == fl.FB_EINVAL
    assert b"references vertex 4" in L.fb_last_error()
    assert L.fb_cut_create(C.byref(h), 0, 0, fl.dptr(v), 1, fl.uptr(bad)) == fl.FB_EINVAL
    assert L.fb_cut_create(None, 0, 4, fl.dptr(v), 1, fl.uptr(bad)) == fl.FB_EINVAL
    assert L.fb_cut_face_centroids(None) == fl.FB_EINVAL
    assert L.fb_cut_read(None, 0, None, None) == fl.FB_EINVAL


def test_slab_plan_deals_every_plane_and_layer_once():
    """fembrain_amd.poly.slab_plan: owned planes partition [0, planes), owned cell layers partition [0, planes - 1), every slab
    holds one plane below and two above what it owns (clipped to the grid)"""
    from fembrain_amd.poly import slab_plan
    for planes in (4, 16, 17, 111, 256):
        for world in (1, 2, 3, 8):
            if planes < 2 * world:
                with pytest.raises(ValueError):
                    slab_plan(planes, world, 0)
                continue
            own_p, own_l = [], []
            for r in range(world):
                p0, p1, zf, zc, op, ol = slab_plan(planes, world, r)
                assert op == p1 - p0 >= 2 and zf == max(p0 - 1, 0) and zf + zc - 1 == min(p1 + 1, planes - 1) and zc >= 2
                own_p += list(range(p0, p0 + op))
                own_l += list(range(p0, p0 + ol))
            assert own_p == list(range(planes)) and own_l == list(range(planes - 1))


def test_c_abi_header_is_plain_c99(tmp_path):
    """include/fembrain_hip.h is the drop-in boundary: a C99 compiler must accept it with -pedantic (no C++-isms) and a C
    program must link against the library"""
    import subprocess
    src = tmp_path / "cabi.c"
    src.write_text('#include "fembrain_hip.h"\n#include "fembrain_hip_testing.h"\n'
                   'int main(void) { fb_fem_params p; fb_fem_default_params(&p); return (p.linear == 0 && fb_last_error() != 0) ? 0 : 1; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "cabi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", os.path.join(root, "fembrain_amd"), "-lfembrain_hip", "-Wl,-rpath," + os.path.join(root, "fembrain_amd")])
    assert subprocess.call([str(exe)]) == 0


# ---- host-staged test communicator: stale segments and lost peers (no device call involved) ----------------------------------
def _comm_worker(rank, world, name, q, behave):
    import ctypes as C
    os.environ["FEMBRAIN_LOCAL_TIMEOUT_MS"] = "1500"
    from fembrain_amd import lib as fl
    L = fl.lib()
    comm = C.c_void_p()
    rc = L.fb_comm_create_local(C.byref(comm), rank, world, name.encode(), 4096, 0)
    if rc != 0:
        q.put((rank, "create", rc))
        return
    mine = np.array([rank + 10], np.int64)
    got = np.zeros(world, np.int64)
    rcs = []
    for k in range(3):
        if behave == "leave" and rank == world - 1 and k == 1:
            q.put((rank, "left", 0))
            q.close()
            q.join_thread()   # flush the feeder thread
            os._exit(0)       # between two collectives, without telling anyone
        rcs.append(L.fb_comm_test_allgather(comm, mine.ctypes.data, got.ctypes.data, 8))
        if rcs[-1] != 0:
            break
    q.put((rank, rcs, got.tolist()))
    L.fb_comm_destroy(comm)


def _run_comm(world, name, behave):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_comm_worker, args=(r, world, name, q, behave)) for r in range(world)]
    for p in ps:
        p.start()
    out = [q.get(timeout=60) for _ in range(world)]
    for p in ps:
        p.join(timeout=20)
        assert not p.is_alive()
    return {o[0]: o[1:] for o in out}


def test_local_communicator_replaces_a_stale_segment():
    """a segment of the same name left by a killed run (non-zero barrier state) must not deadlock the next run"""
    name = "/fembrain_hosttest_%d_stale" % os.getpid()
    with open("/dev/shm" + name, "wb") as f:
        f.write(b"\x07" * 65536)   # magic / count / sense all garbage
    res = _run_comm(3, name, "ok")
    for r in range(3):
        assert res[r][0] == [0, 0, 0] and res[r][1] == [10, 11, 12], res
    assert not os.path.exists("/dev/shm" + name)   # rank 0 unlinked it on destroy


def test_local_communicator_bounds_its_waits_when_a_peer_leaves():
    name = "/fembrain_hosttest_%d_leave" % os.getpid()
    t0 = time.time()
    res = _run_comm(3, name, "leave")
    assert res[2][0] == "left"
    for r in (0, 1):   # the survivors got FB_ECOMM (-5) at the second collective instead of waiting for ever
        assert res[r][0][0] == 0 and res[r][0][-1] == -5, res
    assert time.time() - t0 < 30
    try:
        os.unlink("/dev/shm" + name)
    except FileNotFoundError:
        pass


def test_slab_order_finds_a_grid_again_and_gives_every_rank_two_neighbours():
    """The handle's internal node order (fembrain_amd/csrc/renumber.h, host restatement fb_plan_slab_order): a grid in ANY
    caller order gets its plane-by-plane order back (longest axis first), the widest element shrinks from ~the whole list to one plane,
    and contiguous ranges of the new order are slabs -- at most two neighbour ranks each, where equal ranges of the caller's ids make
    every rank a neighbour of every other (the reference appends cut nodes at the end of the list, VolMesh.cpp:1086-1091)."""
    L = fl.lib()
    v0, t0 = truth_cube(9, 14, 11, 0.1)
    rng = np.random.default_rng(3)
    m = rng.permutation(len(v0))
    v = np.empty_like(v0)
    v[m] = v0
    t = np.ascontiguousarray(m[t0].astype(np.int32))
    o = np.empty(len(v), np.int32)
    a, b = C.c_int(0), C.c_int(0)
    fl.check(L.fb_plan_slab_order(len(v), fl.dptr(v), len(t), fl.iptr(t), fl.iptr(o), C.byref(a), C.byref(b)))
    assert sorted(o.tolist()) == list(range(len(v)))
    # y (14 planes) is the longest axis, then z, then x
    assert np.array_equal(v[o], v0[np.lexsort((v0[:, 0], v0[:, 2], v0[:, 1]))])
    assert a.value > len(v) // 2 and b.value <= 9 * 11 + 11 + 1
    new_of_old = np.empty(len(v), np.int64)
    new_of_old[o] = np.arange(len(v))
    t_int = np.ascontiguousarray(new_of_old[t].astype(np.int32))
    for world in (2, 4, 7):
        for tets, want_few in ((t, False), (t_int, True)):
            worst = 0
            for rank in range(world):
                info, get, (_, h) = _plan(v, tets, [], world, rank)
                worst = max(worst, int((np.diff(get("halo_off")) > 0).sum()))
                L.fb_plan_destroy(h)
            assert (worst <= 2) if want_few else (worst == world - 1), (world, want_few, worst)
            # what a rank votes with under FB_RENUMBER_AUTO (fem.hip vote_shard_order) is the neighbour count of its plan
            for rank in range(world):
                info, get, (_, h) = _plan(v, tets, [], world, rank)
                vote = np.zeros(3, np.int32)
                fl.check(L.fb_plan_shard_vote(len(v), len(tets), fl.iptr(tets), world, rank, None, fl.iptr(vote)))
                assert vote[0] == int((np.diff(get("halo_off")) > 0).sum()) and vote[1] == info["n_tets"] and 0 < vote[2] <= vote[1]
                assert (2 * vote[2] > vote[1]) != want_few or world == 7    # most elements reach into another rank <=> scrambled (7 ranks: thin slabs)
                L.fb_plan_destroy(h)
    bad = t.copy()
    bad[5, 2] = len(v)          # (in an element of rank 0 or not: the range check is the builder's, the vote says "unknown")
    vote = np.zeros(3, np.int32)
    ranks_unknown = 0
    for rank in range(4):
        fl.check(L.fb_plan_shard_vote(len(v), len(bad), fl.iptr(bad), 4, rank, None, fl.iptr(vote)))
        ranks_unknown += vote[0] == -1
    assert ranks_unknown >= 1
    fl.check(L.fb_plan_shard_vote(len(v), len(t), fl.iptr(t), 4, 1, fl.iptr(np.array([0, 10, 10, 20, len(v)], np.int32)), fl.iptr(vote)))
    assert vote[0] == -1        # a rank without nodes
    # an already banded order is found again exactly: nothing to gain
    fl.check(L.fb_plan_slab_order(len(v0), fl.dptr(v0), len(t0), fl.iptr(np.ascontiguousarray(t0)), fl.iptr(o), C.byref(a), C.byref(b)))
    assert b.value <= a.value
    # degenerate input: all nodes on one point -> the caller's order stands
    z = np.zeros((5, 3))
    tz = np.array([[0, 1, 2, 3]], np.int32)
    fl.check(L.fb_plan_slab_order(5, fl.dptr(z), 1, fl.iptr(tz), fl.iptr(o[:5].copy()), C.byref(a), C.byref(b)))
    assert a.value == b.value == 3


def test_synthetic_cut_and_apply_delta_describe_the_same_mesh():
    """meshgen.synthetic_cut (the change the delta re-sync tests and probes use): splitting every element that crosses a plane in four on a
    new centroid node keeps the volume and the orientation of every piece, and the delta it reports, applied to the old mesh by
    meshgen.apply_delta (the contract of fb_fem_resync_delta: in-place changes, ordered erasure, appended elements and nodes), is the mesh
    it returns"""
    from fembrain_amd.meshgen import apply_delta, synthetic_cut, truth_cube
    v, t = truth_cube(7, 8, 9, 0.1)

    def vol(vv, tt):
        p = vv[tt]
        return np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0]) / 6.0
    for axis, every, stride in ((0, 3, 1), (1, 0, 1), (2, 1, 2)):
        v2, t2, d = synthetic_cut(v, t, axis=axis, where=0.37, every_changed=every, stride=stride)
        hit = len(d["removed"]) + len(d["changed_ids"])
        assert hit > 0 and len(d["new_xyz"]) == hit and len(d["added"]) == 4 * len(d["removed"]) + 3 * len(d["changed_ids"])
        assert len(v2) == len(v) + hit and len(t2) == len(t) + 3 * hit
        assert np.all(np.diff(d["removed"]) > 0) and np.all(np.diff(d["changed_ids"]) > 0) and not set(d["removed"]) & set(d["changed_ids"])
        va, ta = apply_delta(v, t, d)
        assert np.array_equal(va, v2) and np.array_equal(ta, t2)
        w0, w2 = vol(v, t), vol(v2, t2)
        assert abs(w2.sum() - w0.sum()) < 1e-12 and (np.sign(w2) == np.sign(w0[0])).all() and np.abs(w2).min() > 1e-9
        kept = np.ones(len(t), bool)
        kept[d["removed"]] = False
        kept[d["changed_ids"]] = False
        assert np.array_equal(t2[: kept.sum() + len(d["changed_ids"])][~np.isin(np.nonzero(np.ones(len(t), bool) & ~np.isin(np.arange(len(t)), d["removed"]))[0], d["changed_ids"])], t[kept])


def test_unstructured_mesh_generator_gives_positive_elements_and_uneven_valences():
    """fembrain_amd.meshgen.delaunay_jittered (bench.py's `delaunay606k` leg, tools/probe_numbering.py case d): every element positively
    oriented, every node used, a fixed slab, and the valences of an unstructured mesh -- hull nodes with several times the neighbours of
    an interior node; the same seed gives the same mesh"""
    from fembrain_amd.meshgen import delaunay_jittered
    v, t, f = delaunay_jittered(10)
    assert v.shape == (1000, 3) and t.dtype == np.int32 and t.min() == 0 and t.max() == 999 and len(np.unique(t)) == 1000
    vol = np.einsum("ij,ij->i", v[t[:, 1]] - v[t[:, 0]], np.cross(v[t[:, 2]] - v[t[:, 0]], v[t[:, 3]] - v[t[:, 0]])) / 6
    assert (vol > 0).all() and abs(vol.sum() - ((v.max(0) - v.min(0)).prod())) < 0.2 * vol.sum()
    assert 0 < len(f) < 200 and (v[f, 0] < 0.1).all()
    pairs = np.unique(np.sort(np.concatenate([t[:, [i, j]] for i in range(4) for j in range(i + 1, 4)]), axis=1), axis=0)
    val = np.bincount(pairs.ravel(), minlength=1000)
    assert val.max() >= 2 * np.median(val)
    v2, t2, f2 = delaunay_jittered(10)
    assert np.array_equal(v, v2) and np.array_equal(t, t2) and np.array_equal(f, f2)
