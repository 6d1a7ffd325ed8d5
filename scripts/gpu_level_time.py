"""Per-step cost of the level estimation (EmptyAngle, extended range) on a BASELINE workload (development aid)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
wl = sys.argv[1] if len(sys.argv) > 1 else "dam_break_1m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
scene_f, params_f, _ = WORKLOADS[wl]
scn = scene_f()
P = params_f(level_estimation_method="EmptyAngle", maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002)
pos, mass, vel = sc.init_particles(scn)
lib = ffi.load_product()
g = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary))
g.upload(mass, pos, vel)
p = P.to_ffi()
for _ in range(3): g.step(p)
g.profile_reset(); g.profile_enable(1)
t0 = time.perf_counter()
for _ in range(steps): st = g.step(p)
dt = time.perf_counter() - t0
print(f"{wl} + EmptyAngle level estimation: {dt/steps*1e3:.2f} ms/step (level estimation {st.ms_level_estimation:.2f} ms)")
for name, (launches, ms) in sorted(g.profile_get().items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {name:24s} {launches/steps:7.1f} launches/step  {ms/steps*1e3:9.1f} us/step")
import numpy as np
print("surface particles:", int(g.download("flag_is_fluid_surface").sum()), "classes:", np.bincount(g.download("particle_size_class"), minlength=5))
