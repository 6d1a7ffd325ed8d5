"""Wall-clock cost of a step with EmptyAngle level estimation on dam_break_1m, uninstrumented (gpu_level_time.py profiles)."""
import sys, time
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import torch  # noqa: F401
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
scene_f, params_f, _ = WORKLOADS["dam_break_1m"]
scn = scene_f()
P = params_f(level_estimation_method="EmptyAngle", maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002)
pos, mass, vel = sc.init_particles(scn)
g = ffi.Context(ffi.load_product(), len(mass), sc.boundary_planes(scn.boundary))
g.upload(mass, pos, vel)
p = P.to_ffi()
for _ in range(20):
    g.step(p)
t0 = time.perf_counter()
for _ in range(100):
    st = g.step(p)
print(f"1M + level estimation: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms/step (level part {st.ms_level_estimation:.3f} ms)")
