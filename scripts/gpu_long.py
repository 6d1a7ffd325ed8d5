"""ms/step over a long run in windows (clock ramp / drift check; development aid)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
scene_f, params_f, _ = WORKLOADS["dam_break_1m"]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
lib = ffi.load_product()
g = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary)); g.upload(mass, pos, vel)
p = P.to_ffi()
for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    t0 = time.perf_counter(); its = []
    for _ in range(250):
        st = g.step(p); its.append(st.div_solver.iters + st.density_solver.iters + 2)
    dt = time.perf_counter() - t0
    print(f"steps {w*250:5d}..{w*250+249:5d}: {dt/250*1e3:.3f} ms/step, mean Jacobi iterations/step {np.mean(its):.2f}, us per iteration-equivalent {dt/250*1e6/ (np.mean(its)+3):.1f}")
