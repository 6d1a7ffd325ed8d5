"""Histogram of the neighbour counts (self included, as the reference counts) of configs[1] along the driver window and beyond."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "dam_break_1m"
scene_f, params_f, _ = WORKLOADS[name]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
g = ffi.Context(ffi.load_product(), len(mass), sc.boundary_planes(scn.boundary))
g.upload(mass, pos, vel)
p = P.to_ffi()
for s in range(121):
    g.step(p)
    if s in (0, 5, 10, 15, 20, 24, 60, 120):
        nc = g.download("neighbor_count").astype(np.int64) - 1   # without the particle itself
        h = np.bincount(nc, minlength=40)
        print(f"step {s}: mean {nc.mean():.2f} max {nc.max()}  >12: {(nc > 12).mean():.4f}  >16: {(nc > 16).mean():.4f}  >20: {(nc > 20).mean():.4f}  >24: {(nc > 24).mean():.5f}  hist[8..24] {h[8:25].tolist()}")
