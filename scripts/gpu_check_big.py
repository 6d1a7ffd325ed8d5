import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
wl = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
scene_f, params_f, _ = WORKLOADS[wl]
scn, P = scene_f(), params_f()
if len(sys.argv) > 3: P = P.replace(max_dt=float(sys.argv[3]))
pos, mass, vel = sc.init_particles(scn)
lib = ffi.load_product()
g = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary))
g.upload(mass, pos, vel)
p = P.to_ffi()
for s in range(nsteps):
    t = time.perf_counter(); st = g.step(p); ms = (time.perf_counter() - t) * 1e3
    gi = g.grid()
    if s < 6 or s % 10 == 0: print("step", s, f"{ms:.2f} ms", st.div_solver.iters, st.density_solver.iters, "grid", gi.size_x, gi.size_y, gi.size_x * gi.size_y)
x = g.download("position"); print("x range", x.min(0), x.max(0))
