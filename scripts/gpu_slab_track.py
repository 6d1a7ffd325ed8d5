"""Does a slab group track the single context, or drift from the first steps?  64k dam break, k loopback ranks (re-balancing every
20 steps) beside one context: simulated time, dt, iteration counts and the largest speed every `every` steps.
usage: gpu_slab_track.py [k] [steps] [every] [rebalance_every]"""
import sys
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
import numpy as np
import torch  # noqa: F401
from adaptive_sph_amd import distributed as D, ffi, scene as sc
from adaptive_sph_amd.workloads import WORKLOADS
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
reb = int(sys.argv[4]) if len(sys.argv) > 4 else 20
scene_f, params_f, _ = WORKLOADS["dam_break_64k"]
scn, P = scene_f(), params_f()
pos, mass, vel = sc.init_particles(scn)
planes = sc.boundary_planes(scn.boundary)
lib = ffi.load_product()
p = P.to_ffi()
single = ffi.Context(lib, len(mass), planes)
single.upload(mass, pos, vel)
if k == 0:   # "twin": one more single context whose x coordinates differ by one ulp -- the yardstick of chaotic divergence
    class _G(list):
        pass
    tw = ffi.Context(lib, len(mass), planes)
    pos2 = pos.copy()
    pos2[:, 0] = np.nextafter(pos2[:, 0], np.float32(10.0))
    tw.upload(mass, pos2, vel)
    grp = None
else:
    grp = D.make_loopback_group(lib, pos, mass, vel, planes, k)
    for c in grp:
        c.dist_set_rebalance(reb)
n = len(mass)
its_s = its_g = 0
for s in range(steps):
    a = single.step(p)
    b = tw.step(p) if grp is None else ffi.group_step(grp, p)[0]
    its_s += a.div_solver.iters + a.density_solver.iters
    its_g += b.div_solver.iters + b.density_solver.iters
    if s % every == every - 1 or s < 5:
        vs = single.download("velocity")
        vg = tw.download("velocity") if grp is None else D.gather_by_id(grp, "velocity", n)
        xs = single.download("position")
        xg = tw.download("position") if grp is None else D.gather_by_id(grp, "position", n)
        print(f"step {s + 1:5d}: time {single.time:.6f} / {(tw if grp is None else grp[0]).time:.6f}  dt {a.dt:.3e} / {b.dt:.3e}  iterations so far {its_s} / {its_g}  "
              f"|v|max {np.sqrt((vs ** 2).sum(1)).max():.3f} / {np.sqrt((vg ** 2).sum(1)).max():.3f}  max |dx| {np.abs(xs - xg).max():.2e}", flush=True)
