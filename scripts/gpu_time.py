"""Wall-clock per step of the HIP path on a BASELINE workload (development aid)."""
import os, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np
from adaptive_sph_amd import ffi, scene as sc, build
from adaptive_sph_amd.workloads import WORKLOADS
wl = sys.argv[1] if len(sys.argv) > 1 else "dam_break_1m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
overrides = eval(sys.argv[3]) if len(sys.argv) > 3 else {}     # e.g. "dict(level_estimation_method='EmptyAngle')"
scene_f, params_f, _ = WORKLOADS[wl]
scn, P = scene_f(), params_f(**overrides)
pos, mass, vel = sc.init_particles(scn)
lib = ffi.load_product()
g = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
g.upload(mass, pos, vel)
p = P.to_ffi()
for _ in range(int(os.environ.get('SPH_TIME_WARMUP', '20'))): g.step(p)
t0 = time.perf_counter()
its = []
for _ in range(steps):
    st = g.step(p); its.append((st.div_solver.iters + 1, st.density_solver.iters + 1))
dt = time.perf_counter() - t0
print(f"{wl}: {dt/steps*1e3:.3f} ms/step, {len(mass)*steps/dt/1e6:.1f} M particle-steps/s, mean iters {np.mean(its,0)}, gpu ms {st.ms_simulation_step:.3f} neigh {st.ms_neighborhood:.3f} div {st.ms_div_solver:.3f} dens {st.ms_density_solver:.3f}")
