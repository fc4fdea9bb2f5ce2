"""Generates tests/golden/sphere_tets_c0.1.npz from the reference's own data file
/root/reference/data/models/sphere/implicit_sphere.veg (4,992 vertices / 3,744 tets written by the reference's old
tandem kernel, data/opencl/PolyOldKernels.cl:86-240: 8 un-welded vertices and 6 tets per included cell, cells in
linear order, cellsize 0.1 on the 12^3-point grid of sphere.blob).

Stored (data only): the grid-cell coordinates (ix,iy,iz) of every included cell in file order, the 6x4 local
corner pattern of the tets, and the vertex coordinates of the first and last cell (fp32) as position samples.
Run here once; the .npz is committed, the reference tree is not needed at test time."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from fembrain_amd.meshgen import read_veg

src = "/root/reference/data/models/sphere/implicit_sphere.veg"
v, t = read_veg(src)
assert len(v) == 4992 and len(t) == 3744
cells = v.reshape(-1, 8, 3)
lo = cells[:, 0, :]
ijk = np.rint((lo + 0.5) / 0.1).astype(np.int32)
# corner k of a cell = 4*dx + 2*dy + dz
for k in range(8):
    d = np.array([(k >> 2) & 1, (k >> 1) & 1, k & 1]) * 0.1
    assert np.abs(cells[:, k, :] - (lo + d)).max() < 1e-6
local = (t.reshape(-1, 6, 4) - (np.arange(len(cells)) * 8)[:, None, None]).astype(np.int32)
assert (local == local[0]).all()
np.savez_compressed(os.path.join(os.path.dirname(__file__), "sphere_tets_c0.1.npz"), cell_ijk=ijk, pattern=local[0],
                    first_cell_xyz=cells[0].astype(np.float32), last_cell_xyz=cells[-1].astype(np.float32),
                    grid=np.array([12, 12, 12], np.int32), cellsize=np.float32(0.1), lower=np.array([-0.5, -0.5, -0.5], np.float32))
print("cells", len(cells), "pattern", local[0].tolist())

# ---- marching-cubes tables: hashes only (the arrays themselves are reference source text and stay there) ----
import hashlib
import json
import re

src = open("/root/reference/src/implicit/_CellConfigTable.h").read()
rows = re.findall(r"\{([^{}]*)\}", src[src.index("g_triTableCache[256][16]"):])[:256]
tri = np.array([[255 if int(x) < 0 else int(x) for x in r.replace(" ", "").split(",") if x.strip()] for r in rows], np.uint8)
assert tri.shape == (256, 16)
src = open("/root/reference/src/implicit/_CellConfigTableCompact.cpp").read()
src = src[src.index("g_numVerticesTableCompact[256]"):]
nvert = np.array([int(x) for x in re.findall(r"\d+", src[src.index("{"):src.index("}")])], np.uint8)
assert nvert.shape == (256,) and np.array_equal(nvert, (tri != 255).sum(1))
json.dump({"tri_table_u8_256x16_sha256": hashlib.sha256(tri.tobytes()).hexdigest(),
           "num_vertices_u8_256_sha256": hashlib.sha256(nvert.tobytes()).hexdigest(),
           "total_indices": int(nvert.sum()),
           "note": "-1 entries hashed as 255; source src/implicit/_CellConfigTable.h:58-317 and _CellConfigTableCompact.cpp"},
          open(os.path.join(os.path.dirname(__file__), "mc_table.json"), "w"), indent=1)
print("mc table hashed,", int(nvert.sum()), "indices")
