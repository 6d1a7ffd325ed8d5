"""Per-kernel HIP-event times of configs[1] + EmptyAngle (profiler mode 1), 10 steps after 20: which launches the level estimation costs.
usage: [SPH_HIP_LIBRARY=libsph_lab.so SPH_LEVEL_TILES=k] python scripts/gpu_level_kernels.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import dam_break_params
scn = sc.dam_break_1m()
pos, mass, vel = sc.init_particles(scn)
g = ffi.Context(ffi.load_product(), len(mass), sc.boundary_planes(scn.boundary))
g.upload(mass, pos, vel)
p = dam_break_params(level_estimation_method="EmptyAngle", maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002).to_ffi()
for _ in range(20):
    g.step(p)
g.profile_reset()
g.profile_enable(1)
for _ in range(10):
    g.step(p)
for k, (n, ms) in sorted(g.profile_get().items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} {n / 10:7.1f} launches/step  {ms * 1e3 / max(n, 1):8.2f} us each  {ms / 10:7.3f} ms/step")
