"""Level estimation on a slab decomposition, ONE GPU: k loopback ranks stepping configs[1] with the EmptyAngle detector, the
propagation in the frontier form with probing halo members (default) or with every unassigned particle in every sweep
(SPH_SLAB_LEVEL_PLAIN=1), against the single context.  usage: gpu_slab_level_time.py [k] [steps]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS

k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scene_f, params_f, _ = WORKLOADS["dam_break_1m"]
scn = scene_f()
P = params_f(level_estimation_method='EmptyAngle', maximum_surface_distance=0.2, particle_radius_fine=0.0005, particle_radius_base=0.002)
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary)
lib = ffi.load_product()
p = P.to_ffi()
grp = D.make_loopback_group(lib, pos, mass, vel, planes, k)
for w in range(10):
    ffi.group_step(grp, p)
[c.dist_get_stats(reset=True) for c in grp]
t0 = time.perf_counter()
for _ in range(steps):
    sts = ffi.group_step(grp, p)
tk = (time.perf_counter() - t0) / steps * 1e3
st = grp[0].dist_get_stats()
print(f"{k} loopback ranks, level estimation {'plain' if os.environ.get('SPH_SLAB_LEVEL_PLAIN') else 'frontier'}: {tk:.3f} ms/step, "
      f"level estimation {sts[0].ms_level_estimation:.3f} ms (host clock, last step), exchanges per step {st['exchanges'] / steps:.1f}")
lv = np.concatenate([c.download('level_estimation') for c in grp]) if hasattr(grp[0], 'download') else None
print("level checksum", float(np.nansum(lv)) if lv is not None else None, "assigned", int(np.isfinite(lv).sum()) if lv is not None else None)
