"""k slab ranks of one process on ONE GPU (loopback transport) stepping a weak-scaling scene: functional check of the
decomposition at full size (development aid; says nothing about multi-GPU speed)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.distributed import make_loopback_group, gather_by_id
from adaptive_sph_amd.workloads import WORKLOADS
wl, k, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
scene_f, params_f, _ = WORKLOADS[wl]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
lib = ffi.load_product()
planes = sc.boundary_planes(scn.boundary)
ctxs = make_loopback_group(lib, pos, mass, vel, planes, k)
p = P.to_ffi()
t0 = time.perf_counter()
for s in range(steps):
    sts = ffi.group_step(ctxs, p)
dt = time.perf_counter() - t0
print(f"{wl} x{k} loopback: {dt/steps*1e3:.2f} ms/step, owned per rank {[c.n for c in ctxs]}, dt {sts[0].dt:.3e}, iters {sts[0].div_solver.iters} {sts[0].density_solver.iters}")
one = ffi.Context(lib, len(mass), planes); one.upload(mass, pos, vel)
for s in range(steps): one.step(p)
for f in ["position", "velocity", "density"]:
    a, b = gather_by_id(ctxs, f, len(mass)), one.download(f)
    print(f, "max rel diff vs single context:", float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)))
