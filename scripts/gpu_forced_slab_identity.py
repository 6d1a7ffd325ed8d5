"""ONE rank in forced slab mode against the plain context over many steps of configs[1]: no ghosts, no migrants -- the slab driver's
own bookkeeping (fused refresh, appended arrays, keys behind the last cell, ownership flags, all-reduced decisions) must not change a bit.
usage: gpu_forced_slab_identity.py [steps]"""
import os, sys, ctypes as C
os.environ["SPH_FORCE_SLAB_MODE"] = "1"
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import numpy as np
import torch  # noqa: F401
from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
scene_f, params_f, _ = WORKLOADS["dam_break_1m"]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
lib = ffi.load_product()
raw = (C.c_uint8 * 128)()
lib.comm_unique_id(raw)
planes = sc.boundary_planes(scn.boundary)
c = ffi.Context(lib, len(mass) + 65536, planes)
c.dist_configure(0, 1, -D.INF, D.INF)
c.comm_init(bytes(raw), 0, 1)
del os.environ["SPH_FORCE_SLAB_MODE"]
c.upload(mass, pos, vel)
c.upload_field("particle_id", np.arange(len(mass), dtype=np.uint32))
g = ffi.Context(lib, len(mass), planes)
g.upload(mass, pos, vel)
p = P.to_ffi()
for s in range(steps):
    a, b = g.step(p), c.step(p)
    assert a.dt == b.dt and a.div_solver.iters == b.div_solver.iters and a.density_solver.iters == b.density_solver.iters, s
    if s % 100 == 99 or s == steps - 1:
        ids = c.download("particle_id")
        bad = [f for f in ("position", "velocity", "density", "pressure") if not np.array_equal(D.gather_by_id([c], f, len(mass)), g.download(f))]
        print(f"step {s + 1}: ids complete {np.array_equal(np.sort(ids), np.arange(len(mass)))}, fields that differ: {bad or 'none'}", flush=True)
        assert not bad
print("bit-identical over", steps, "steps")
