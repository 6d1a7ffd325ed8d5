"""Two builds of the library in ONE process, the same free-running steps: are the results bit for bit the same?
usage: python scripts/gpu_compare_libs.py libA.so libB.so [workload=dam_break_1m] [steps=25]     (files in adaptive_sph_amd/csrc/)
(both are linked -Bsymbolic and loaded RTLD_LOCAL: neither interposes the other's symbols)"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402
from adaptive_sph_amd import ffi, scene as sc  # noqa: E402
from adaptive_sph_amd.workloads import WORKLOADS  # noqa: E402

a, b = sys.argv[1], sys.argv[2]
wl = sys.argv[3] if len(sys.argv) > 3 else "dam_break_1m"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 25
libs = [ffi.SphLibrary(ffi.PKG_DIR / "csrc" / n, "sph_", global_symbols=False) for n in (a, b)]
scene_f, params_f, _ = WORKLOADS[wl]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
ctx = [ffi.Context(l, len(mass), planes) for l in libs]
for c in ctx:
    c.upload(mass, pos, vel)
p = P.to_ffi()
for s in range(steps):
    st = [c.step(p) for c in ctx]
    k = [(float(x.dt), int(x.div_solver.iters), int(x.density_solver.iters), int(x.div_solver.normal_count), int(x.density_solver.normal_count)) for x in st]
    if k[0] != k[1]:
        print(f"step {s}: {k[0]} != {k[1]}")
        break
bad = [f for f in ("position", "velocity", "density", "pressure", "pressure_accel", "aii", "ppe_source_term", "constant_field", "neighbor_count")
       if not np.array_equal(ctx[0].download(f), ctx[1].download(f))]
print(f"{wl}, {steps} steps, {a} vs {b}:", "BIT-IDENTICAL" if not bad else f"DIFFERENT in {bad}")
