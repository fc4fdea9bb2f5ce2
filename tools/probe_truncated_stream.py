"""Development aid (WRONG RESULTS by construction): what an iteration of k_pcg_pipe costs at the 56^3 cube when every slice streams n slots
fewer (FEMBRAIN_PIPE_TRUNCATE=n in the helpers' task table, FEMBRAIN_PIPE_HELPERS=2: the table without helpers).  The solve does not
converge to anything meaningful; it runs to the iteration cap and the time per iteration is what is read -- the price of a streamed slot,
for the half-storage estimate in DESIGN.md section 4.  usage: probe_truncated_stream.py [n=56]"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
n = int(sys.argv[1])
v, t = truth_cube(n, n, n, 0.1)
fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
g = FemIntegrator(v, t, fixed)
g.set_uniform_force(1, -10000.0)
g.system()
it = 500
us = [round(g.time_persist(3, it) / it * 1e6, 2) for _ in range(3)]      # (launches of 500 iterations each, whatever the iterates are)
print(json.dumps(dict(truncate=os.environ.get("FEMBRAIN_PIPE_TRUNCATE", "-"), helpers_table=os.environ.get("FEMBRAIN_PIPE_HELPERS", "-"), kernel=g.pcg_path()["kernel"], iterations=it, us_per_iteration=us)))
''' % ROOT
n = sys.argv[1] if len(sys.argv) > 1 else "56"
for tr in (None, "0", "1", "2", "3", "4", "6"):
    env = dict(os.environ)
    if tr is not None:
        env["FEMBRAIN_PIPE_HELPERS"] = "2"
        env["FEMBRAIN_PIPE_TRUNCATE"] = tr
    out = subprocess.run([sys.executable, "-c", CHILD, n], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout.strip() or out.stderr[-800:], flush=True)
