"""Where a step's wall time goes outside the C step: ctypes call vs the step's own wall clock (development aid)."""
import os, sys, time, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
scene_f, params_f, _ = WORKLOADS["dam_break_1m"]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
lib = ffi.load_product()
g = ffi.Context(lib, len(mass), sc.boundary_planes(scn.boundary, P.init_boundary_handler))
g.upload(mass, pos, vel)
p = P.to_ffi()
for _ in range(30): g.step(p)
st = ffi.SphStepStats()
fn = lib.step
h = g.handle
pp, ps = C.byref(p), C.byref(st)
inner = []
t0 = time.perf_counter()
for _ in range(200):
    fn(h, pp, ps)
    inner.append(st.ms_simulation_step)
dt = (time.perf_counter() - t0) / 200 * 1e3
print(f"raw ctypes loop: {dt:.3f} ms/step wall; the step's own clock (group_step_inner start -> end): mean {np.mean(inner):.3f} ms, median {np.median(inner):.3f}")
hip = C.CDLL("libamdhip64.so")
t0 = time.perf_counter()
for _ in range(2000): hip.hipSetDevice(0)
print(f"hipSetDevice: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per call")
