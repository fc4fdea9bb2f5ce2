"""Quick performance probe of the FEM path on a truth cube (development aid)."""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube

n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
prec = fl.FB_MATRIX_F64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else fl.FB_MATRIX_F32
variant = int(sys.argv[4]) if len(sys.argv) > 4 else fl.FB_PCG_MERGED
t0 = time.time()
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
t1 = time.time()
g = FemIntegrator(v, t, fixed, matrix_precision=prec, pcg_variant=variant)
t2 = time.time()
print("mesh %d nodes %d tets: gen %.2fs create %.2fs blocks %d" % (len(v), len(t), t1 - t0, t2 - t1, g.num_blocks()), flush=True)
g.set_uniform_force(1, -10000.0)
for k in range(steps):
    ts = time.time()
    it = g.do_timestep()
    te = time.time()
    print("step %d: wall %.4fs iters %d assembly %.3f ms solve %.3f ms -> %.2f us/iter" %
          (k, te - ts, it, g.last.assembly_seconds * 1e3, g.last.solve_seconds * 1e3, g.last.solve_seconds * 1e6 / max(it, 1)), flush=True)
sp = g.time_spmv(200)
by = g.spmv_bytes()
print("spmv %.2f us, %.1f MB algorithmic -> %.1f GB/s" % (sp * 1e6, by / 1e6, by / sp / 1e9))
asm = g.time_assembly(20)
ab = g.assembly_bytes()
print("assembly %.2f us, %.1f MB -> %.1f GB/s" % (asm * 1e6, ab / 1e6, ab / asm / 1e9))
q, _, _ = g.get_q_state()
print("|q|2 = %.6f max|q| = %.6f" % (np.linalg.norm(q), np.abs(q).max()))
