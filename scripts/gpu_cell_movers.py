"""How many particles change their grid cell from one step to the next (what an incremental re-sort would have to move).
usage: python scripts/gpu_cell_movers.py [workload] [steps]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "dam_break_1m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 125
scene_f, params_f, _ = WORKLOADS[name]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
g = ffi.Context(ffi.load_product(), len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
g.upload(mass, pos, vel)
p = P.to_ffi()
prev = None
for s in range(steps):
    g.step(p)
    cs = np.float32(g.grid().cell_size)
    x = g.download("position")
    cell = np.floor(x / cs).astype(np.int64)   # (f32 division, as the cell index is computed)
    if prev is not None:
        moved = int((cell != prev).any(axis=1).sum())
        if s < 30 or s % 10 == 0:
            print(f"step {s}: {moved} of {len(mass)} particles changed cell ({moved / len(mass):.5f})", flush=True)
    prev = cell
