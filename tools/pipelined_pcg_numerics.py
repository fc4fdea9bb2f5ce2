"""Development aid (CPU, numpy): the pipelined Jacobi-PCG recurrences of k_pcg_pipe (Ghysels & Vanroose) against the literal ones on the
oracle's systems of a truth cube -- iteration counts, true residuals, with the exact-residual refresh of every 30th iteration in its full
and partial form and without it.   python tools/pipelined_pcg_numerics.py <nodes per side> [steps] [f32: round the matrix to fp32]"""
import sys, time, numpy as np, scipy.sparse as sp
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd.meshgen import truth_cube, cube_fixed_plane_i0, fixed_vertices_to_dofs
from oracle.pyoracle import OrcFem
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
f = np.zeros(3 * len(v)); f[1::3] = -10000.0
o = OrcFem(v, t); o.integrator(fixed)
free = np.setdiff1d(np.arange(3 * len(v)), fixed)

def cg(A, b, iv, eps=1e-6, maxit=10000):
    x = np.zeros_like(b); r = b - A @ x; d = iv * r; rn = np.sum(r * r * iv); rn0 = rn; it = 1
    while rn > eps * eps * rn0 and it <= maxit:
        q = A @ d; a = rn / (d @ q); x += a * d
        if it % 30 == 0: r = b - A @ x
        else: r -= a * q
        rn_new = np.sum(r * r * iv); d = iv * r + (rn_new / rn) * d; rn = rn_new; it += 1
    return it - 1, x

def pipecg(A, b, iv, eps=1e-6, maxit=10000, refresh=30, full=True):
    # Ghysels-Vanroose pipelined PCG, Jacobi M folded: u = iv r, m = iv w, q = iv s
    x = np.zeros_like(b); r = b - A @ x; w = A @ (iv * r)
    z = np.zeros_like(b); s = np.zeros_like(b); p = np.zeros_like(b)
    gam_old = 1.0; alpha_old = 1.0; it = 1
    gam0 = None
    while it <= maxit:
        u = iv * r
        gam = r @ u; delta = w @ u
        if gam0 is None: gam0 = gam
        if not (gam > eps * eps * gam0): break
        nvec = A @ (iv * w)
        if it > 1:
            beta = gam / gam_old; alpha = gam / (delta - beta * gam / alpha_old)
        else:
            beta = 0.0; alpha = gam / delta
        z = nvec + beta * z; s = w + beta * s; p = u + beta * p
        x += alpha * p
        if it % refresh == 0:
            r = b - A @ x; w = A @ (iv * r)
            if full: s = A @ p; z = A @ (iv * s)
        else:
            r -= alpha * s; w -= alpha * z
        gam_old = gam; alpha_old = alpha; it += 1
    return it - 1, x

for st in range(steps):
    o.set_external_forces(f)
    o.step_prepare()
    ia, ja, a = o.sys_csr()
    A = sp.csr_matrix((a, ja, ia))
    if len(sys.argv) > 3: A = sp.csr_matrix((a.astype(np.float32).astype(np.float64), ja, ia))
    full_rhs = None
    info, keff, rhs, dv = o.step(want=True)
    b = rhs[free]
    iv = 1.0 / A.diagonal()
    t0 = time.time(); i1, x1 = cg(A, b, iv); t1 = time.time(); i2, x2 = pipecg(A, b, iv); t2 = time.time(); i3, x3 = pipecg(A, b, iv, full=False); i4, x4 = pipecg(A, b, iv, refresh=10**9); print('partial-refresh', i3, 'true res %.3e' % (np.linalg.norm(b - A @ x3) / np.linalg.norm(b)), '| no-refresh', i4, 'true res %.3e' % (np.linalg.norm(b - A @ x4) / np.linalg.norm(b)))
    print("n=%d step %d oracle iters %d | cg %d (%.1fs) | pipecg %d (%.1fs) | rel diff x %.3e | true res cg %.3e pipe %.3e" % (
        n, st, abs(info), i1, t1 - t0, i2, t2 - t1, np.abs(x1 - x2).max() / np.abs(x1).max(),
        np.linalg.norm(b - A @ x1) / np.linalg.norm(b), np.linalg.norm(b - A @ x2) / np.linalg.norm(b)))
