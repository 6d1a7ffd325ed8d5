"""level estimation on the side stream vs serial (SPH_LEVEL_SERIAL=1 in the environment selects serial): timing + a checksum of the fields"""
import sys, time, zlib
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
wl = sys.argv[1] if len(sys.argv) > 1 else "dam_break_1m"
scene_f, params_f, _ = WORKLOADS[wl]
kw = dict(level_estimation_method='EmptyAngle', maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002) if wl == "dam_break_1m" else dict(level_estimation_method='EmptyAngle')
scn, P = scene_f(), params_f(**kw)
pos, mass, vel = sc.init_particles(scn)
g = ffi.Context(ffi.load_product(), len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
g.upload(mass, pos, vel)
p = P.to_ffi()
for _ in range(10): g.step(p)
t0 = time.perf_counter()
for _ in range(30): st = g.step(p)
dt = (time.perf_counter() - t0) / 30
crc = [zlib.crc32(g.download(f).tobytes()) for f in ("level_estimation", "position", "flag_is_fluid_surface", "stash")]
print(f"{wl} + level estimation: {dt*1e3:.3f} ms/step  checksums {crc}")
