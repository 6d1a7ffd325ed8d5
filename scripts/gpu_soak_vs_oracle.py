"""Does the CPU oracle fail where the device does?  A free-running configuration on both sides, each until its first error
(development aid: tells a physical blow-up of the scheme from a defect).  usage: gpu_soak_vs_oracle.py workload steps "dict(...)" """
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
from tests.oracle_harness import load_oracle
wl, steps, ov = sys.argv[1], int(sys.argv[2]), eval(sys.argv[3])
scene_f, params_f, _ = WORKLOADS[wl]
scn, P = scene_f(), params_f(**ov)
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
p = P.to_ffi()
for name, lib in (("device", ffi.load_product()), ("oracle", load_oracle())):
    c = ffi.Context(lib, len(mass), planes)
    c.upload(mass, pos, vel)
    last = None
    for s in range(steps):
        try:
            st = c.step(p)
        except ffi.SphError as e:
            print(f"{name}: FAILED at step {s}: {e}  (last ok: time {last[0]:.4f} dt {last[1]:.3e} iters {last[2]})", flush=True)
            break
        last = (st.time, st.dt, (int(st.div_solver.iters), int(st.density_solver.iters)))
        if s % 100 == 0:
            rho = c.download("density")
            print(f"{name}: step {s} time {st.time:.4f} dt {st.dt:.3e} iters {last[2]} rho [{rho.min():.3f}, {rho.max():.3f}]", flush=True)
    else:
        print(f"{name}: {steps} steps ok, time {last[0]:.4f}", flush=True)
