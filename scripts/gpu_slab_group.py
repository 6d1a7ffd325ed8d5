"""k loopback ranks stepping dam_break_1m (for kernel traces of the slab machinery).  usage: gpu_slab_group.py k steps"""
import sys, time
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import torch  # noqa: F401
from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
k, steps = int(sys.argv[1]), int(sys.argv[2])
scene_f, params_f, _ = WORKLOADS["dam_break_1m"]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
lib = ffi.load_product()
grp = D.make_loopback_group(lib, pos, mass, vel, sc.boundary_planes(scn.boundary), k)
p = P.to_ffi()
t0 = time.perf_counter()
for _ in range(steps):
    ffi.group_step(grp, p)
print(f"{k} ranks: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
