"""A long free-running run of one workload on two contexts side by side -- the product's defaults and SPH_INC_SORT=0 (or another switch) --
compared bit for bit every few hundred steps: does the incremental merge (grids that travel, epochs, the mover threshold, builds that
are not adopted) ever leave the radix sort's order?   usage: gpu_long_identity.py <workload> <steps> [every=200] [SWITCH=value] [ranks]
ranks > 1: two loopback slab groups of that many ranks instead of two plain contexts (the slab ranks' merge of their arrivals)."""
import os
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from adaptive_sph_amd import ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS

wl, steps = sys.argv[1], int(sys.argv[2])
every = int(sys.argv[3]) if len(sys.argv) > 3 else 200
sw, val = (sys.argv[4].split("=") if len(sys.argv) > 4 else ("SPH_INC_SORT", "0"))
scene_f, params_f, _ = WORKLOADS[wl]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary, P.init_boundary_handler)
lib = ffi.load_product()
ranks = int(sys.argv[5]) if len(sys.argv) > 5 else 1
if ranks > 1:
    from adaptive_sph_amd import distributed as D

    class Group:   # (the few calls this script makes, on a loopback group)
        def __init__(self):
            self.g = D.make_loopback_group(lib, pos, mass, vel, planes, ranks)
            for c in self.g:
                c.profile_enable(1)

        def step(self, p):
            return ffi.group_step(self.g, p)[0]

        def download(self, f):
            return D.gather_by_id(self.g, f, len(mass))

        def profile_get(self):
            pr = self.g[0].profile_get()
            return {"inc_reorder": pr.get("inc_place", (0, 0)), "sort_scatter": pr.get("sort_scatter", (0, 0))}

    a = Group()
    os.environ[sw] = val
    b = Group()
    del os.environ[sw]
else:
    a = ffi.Context(lib, len(mass), planes)
    os.environ[sw] = val
    b = ffi.Context(lib, len(mass), planes)
    del os.environ[sw]
    for g in (a, b):
        g.upload(mass, pos, vel)
        g.profile_enable(1)
p = P.to_ffi()
t0 = time.perf_counter()
bad = 0
for s in range(steps):
    try:
        sa, sb = a.step(p), b.step(p)
    except ffi.SphError as e:
        print(f"step {s}: {str(e)[:100]} (a guard of the scheme: the run ends here)")
        break
    if (int(sa.div_solver.iters), int(sa.density_solver.iters), float(sa.dt)) != (int(sb.div_solver.iters), int(sb.density_solver.iters), float(sb.dt)):
        print(f"step {s}: statistics differ")
        bad += 1
        break
    if s % every == every - 1 or s == steps - 1:
        same = all(np.array_equal(a.download(f), b.download(f)) for f in ("position", "velocity", "pressure", "density", "neighbor_count"))
        pa = a.profile_get()
        print(f"step {s}: {'identical' if same else 'DIFFERENT'}; default context so far: merges {pa.get('inc_reorder', (0, 0))[0]}, radix scatters "
              f"{pa.get('sort_scatter', (0, 0))[0]}; dt {float(sa.dt):.3g}, {(time.perf_counter() - t0):.1f} s", flush=True)
        bad += 0 if same else 1
        if not same:
            break
print("differences:", bad)
sys.exit(1 if bad else 0)
